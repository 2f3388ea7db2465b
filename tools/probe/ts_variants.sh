#!/bin/bash
# GPU box: TS in both modes for each library given (paths relative to the repo root; "-" = the in-tree build)
for lib in "$@"; do
  for mode in 0 1; do
    if [ "$lib" = "-" ]; then unset GLOME_DEBUG_LIB; else export GLOME_DEBUG_LIB=$PWD/$lib; fi
    timeout -k 10 200 python bench.py --scene TS --mode $mode --no-cpu > gpurun_out/var_TS$mode.json 2> gpurun_out/var_TS$mode.err || { tail -5 gpurun_out/var_TS$mode.err; exit 1; }
    python -c "
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'mode', sys.argv[3], j['ms_per_step'], 'ms; single frame', j['latency']['single_frame_ms'])" gpurun_out/var_TS$mode.json "$lib" $mode
  done
done
