#!/bin/bash
# ms per step of SHORT runs (the driver's --steps 20 --warmup 5) by launches in flight x frames per launch
out=gpurun_out/short_run_sweep.log; : > $out
for steps in ${STEPS:-20 200}; do
for lanes in ${LANES:-2 3 4 6}; do for group in ${GROUPS_:-1 2 4 5 7 8 10 16}; do
  timeout -k 10 120 python bench.py --steps $steps --warmup 5 --no-cpu --lanes $lanes --group $group 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'steps': $steps, 'lanes': $lanes, 'group': $group, 'ms_per_step': j['ms_per_step']}))" | tee -a $out
done; done; done
