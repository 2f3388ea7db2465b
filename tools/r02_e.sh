#!/bin/bash
out=gpurun_out/r02e; mkdir -p $out
NF=1,4,8 PERCU=0,16,24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | tee -a $out/lone.log
for cfg in "4 4" "1 8" "2 8" "4 1" "4 8"; do set -- $cfg; timeout -k 10 120 python bench.py --no-cpu --lanes $1 --group $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined lanes $1 group $2', j['ms_per_step'], j['latency']['single_frame_ms'], j['latency']['ms_per_frame_in_a_lone_launch'])" | tee -a $out/lone.log; done
