#!/bin/bash
out=gpurun_out/r02c; mkdir -p $out
NF=1,2,4,8 PERCU=0,8,16,24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | tee -a $out/lone.log
timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined 4x4', j['ms_per_step'], j['latency'])" | tee -a $out/lone.log
for cfg in "1 8" "2 8" "2 4" "4 2" "4 8" "8 1"; do set -- $cfg; timeout -k 10 120 python bench.py --no-cpu --lanes $1 --group $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined lanes $1 group $2', j['ms_per_step'])" | tee -a $out/lone.log; done
rm -f $out/parity.log
GLOME_PARITY_LOG=$(pwd)/$out/parity.log timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
