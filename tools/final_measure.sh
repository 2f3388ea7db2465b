#!/bin/bash
# The measurement set committed under profiles/ at the end of a round: every configuration, the default bench line, the
# rocprofv3 kernel stats of the same command, the PMC passes, the multi-GPU rehearsal.  Run on the GPU box.
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out gpurun_out/configs
bash tools/run_configs.sh > $out/configs.log 2>&1
cp gpurun_out/configs/*.json $out/ 2>/dev/null
timeout -k 10 300 python bench.py > $out/bench_S3.json 2> $out/bench_S3.err || exit 1
tools/pmc_pass.sh $out/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" || exit 1
python tools/pmc_summary.py $out/pmc > $out/pmc_summary.json
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python tools/shard_timing.py > $out/shard_timing.log 2>&1
GPU_MAX_HW_QUEUES=8 WORK_TILES=64,64 timeout -k 10 600 python tools/shard_balance.py > $out/shard_balance.log 2>&1
timeout -k 10 300 python tools/bih_build_timing.py > $out/bih_build.log 2>&1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/prof --output-format csv -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu > $root/$out/prof.log 2>&1
cd $root
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_S3_kernel_stats.csv
tail -10 $out/configs.log | cut -c1-200
head -3 $out/bench_S3_kernel_stats.csv
python - <<PY
import json
j=json.loads(open("$out/bench_S3.json").read().strip().splitlines()[-1])
print("bench", j["ms_per_step"], j["value"], j["fps"], j["roofline"]["kernel_ms_avg"], j["roofline"]["launches_timed"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["cpu_baseline"]["gpu_over_cpu"], j["cpu_baseline"]["value_1_thread"])
d=json.load(open("$out/pmc_summary.json"))
for k,v in d.items(): print(k[:70], {c:round(x["mean"],1) for c,x in v.items()})
PY
grep "\"world\": [1248], \"launches_in_flight\": 4" $out/shard_timing.log
