#!/bin/bash
# A/B of library variants (tools/build_variants.py) on the GPU box, each in bench.py's default invocation (200 steps) AND in the
# driver's (--steps 20 --warmup 5), alternating, two rounds, inside ONE gpurun call.   usage: tools/ab_short_long.sh "v1 v2 base" [scene] [mode]
variants=$1; scene=${2:-S3}; mode=${3:-0}
for round in 1 2; do for v in $variants; do
  if [ $v = base ]; then unset GLOME_DEBUG_LIB; else export GLOME_DEBUG_LIB=glome_amd/variants/$v.so; fi
  for inv in "" "--steps 20 --warmup 5"; do
    timeout -k 10 300 python bench.py --scene $scene --mode $mode --no-cpu $inv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $scene mode $mode steps', j['steps'], 'orbit', j['ms_per_step'], 'fixed', (j.get('fixed_camera') or {}).get('ms_per_step'), 'single', j['latency']['single_frame_ms'], 'lone8', j['latency']['ms_per_frame_in_a_lone_launch'], flush=True)" || exit 1
  done
done; done
