#!/bin/bash
export PYTHONUNBUFFERED=1
for i in 1 2 3 4 5; do
  timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "full_size or s5 or two_row or golden" > gpurun_out/rep_$i.log 2>&1; tail -1 gpurun_out/rep_$i.log | sed "s/^/new $i: /"
  grep -q failed gpurun_out/rep_$i.log && { grep -n "assert\|Error\|^E " gpurun_out/rep_$i.log | head -20; }
done
