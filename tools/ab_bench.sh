#!/bin/bash
# A/B of two builds of the library on the GPU box: build_old/libglome_hip_old.so (GLOME_DEBUG_LIB) against the in-tree one
# usage: tools/ab_bench.sh "S3 0" "S5 0" ...
for spec in "$@"; do set -- $spec
  for which in old new old new; do
    if [ $which = old ]; then export GLOME_DEBUG_LIB=build_old/libglome_hip_old.so; else unset GLOME_DEBUG_LIB; fi
    timeout -k 10 300 python bench.py --scene $1 --mode $2 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $1 mode $2', j['ms_per_step'], j['value'], 'single', j['latency']['single_frame_ms'], 'lone', j['latency']['ms_per_frame_in_a_lone_launch'])" || exit 1
  done
done
