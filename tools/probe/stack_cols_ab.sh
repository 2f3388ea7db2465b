#!/bin/bash
for spec in "S5mesh 0" "S3mesh 0" "S2 0" "S1 0" "S3 0" "S5 0" "S4 0" "TS 0" "S3 1"; do set -- $spec
  for which in nocols base nocols base; do
    if [ $which = base ]; then unset GLOME_DEBUG_LIB; else export GLOME_DEBUG_LIB=glome_amd/variants/$which.so; fi
    timeout -k 10 300 python bench.py --scene $1 --mode $2 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $1 mode $2', j['ms_per_step'], j['value'], 'single', j['latency']['single_frame_ms'], 'lone', j['latency']['ms_per_frame_in_a_lone_launch'], flush=True)" || exit 1
  done
done
