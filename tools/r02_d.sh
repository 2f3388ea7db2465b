#!/bin/bash
out=gpurun_out/r02d; mkdir -p $out
for qm in "0 1" "1 1" "2 1" "3 2" "3 4" "3 8" "3 16"; do set -- $qm
  GLOME_DEBUG_QMAP=$1 GLOME_DEBUG_QREGIONS=$2 NF=8 PERCU=24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | sed "s/^/qmap $1 regions $2 /" | tee -a $out/lone.log
  GLOME_DEBUG_QMAP=$1 GLOME_DEBUG_QREGIONS=$2 timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qmap $1 regions $2 pipelined 4x4', j['ms_per_step'])" | tee -a $out/lone.log
done
