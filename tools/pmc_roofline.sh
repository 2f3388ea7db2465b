#!/bin/bash
# usage: tools/pmc_roofline.sh TAG [SCENE] [MODE] [GROUP]   -> gpurun_out/TAG_pmc_SCENE_modeM/ + gpurun_out/TAG_pmc_SCENE_modeM.json
# The counter passes behind bench.py's roofline: rocprofv3 --kernel-trace --pmc, one pass per counter group (the pool refuses
# --pmc together with other trace domains), over the launch shape bench.py times alone (one launch in flight, GROUP frames).
tag=$1; scene=${2:-S3}; mode=${3:-0}; group=${4:-8}
root=$(pwd)
out=$root/gpurun_out/${tag}_pmc_${scene}_mode${mode}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  SCENE=$scene MODE=$mode GROUP=$group LAUNCHES=10 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out/pass$i --output-format csv -- python3 $root/tools/pmc_run.py > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
cd $root
python3 tools/pmc_roofline.py $out $scene $mode $group ${out}.json
