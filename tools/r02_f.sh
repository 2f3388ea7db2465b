#!/bin/bash
out=gpurun_out/r02f; mkdir -p $out
for qc in 1 8 64 512; do
GLOME_DEBUG_QCHUNK=$qc NF=8 PERCU=24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | sed "s/^/chunk $qc /" | tee -a $out/lone.log
for cfg in "4 4" "2 8"; do set -- $cfg; GLOME_DEBUG_QCHUNK=$qc timeout -k 10 120 python bench.py --no-cpu --lanes $1 --group $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $qc pipelined lanes $1 group $2', j['ms_per_step'], j['latency']['single_frame_ms'], j['latency']['ms_per_frame_in_a_lone_launch'])" | tee -a $out/lone.log; done
done
