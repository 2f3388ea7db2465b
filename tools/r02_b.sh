#!/bin/bash
out=gpurun_out/r02b; mkdir -p $out
for il in 0 1 4 16 64 256; do
  GLOME_DEBUG_INTERLEAVE=$il NF=4,8 PERCU=16,24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | tee -a $out/lone.log
done
for il in 0 1 16 64; do
  GLOME_DEBUG_INTERLEAVE=$il timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined interleave', $il, j['ms_per_step'], j['latency'])" | tee -a $out/lone.log
done
rm -f $out/parity.log
GLOME_PARITY_LOG=$(pwd)/$out/parity.log timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
