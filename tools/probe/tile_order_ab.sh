#!/bin/bash
# the order in which a launch works through its tiles (GLOME_DEBUG_TILE_ORDER), in bench.py's default and the driver's invocation
scene=${1:-S3}
for round in 1 2; do for o in default bottomup rowmajor reverse; do
  if [ $o = default ]; then unset GLOME_DEBUG_TILE_ORDER; else export GLOME_DEBUG_TILE_ORDER=$o; fi
  for inv in "" "--steps 20 --warmup 5"; do
    timeout -k 10 300 python bench.py --scene $scene --no-cpu --orbit 0 $inv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o $scene steps', j['steps'], 'ms_per_step', j['ms_per_step'], 'single', j['latency']['single_frame_ms'], 'lone8', j['latency']['ms_per_frame_in_a_lone_launch'], 'same frame', j['frame_equals_single_gpu_render'], flush=True)" || exit 1
  done
done; done
