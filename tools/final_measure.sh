#!/bin/bash
# The measurement set committed under profiles/ at the end of a round.  Run on the GPU box:  tools/final_measure.sh r02_d
tag=${1:-final}
out=gpurun_out/$tag
root=$(pwd)
mkdir -p $out
# (PART=pmc: only step 1; PART=rest: everything else; unset: all -- the counter passes of twelve lines take most of a 20-minute call)
# 1. counter passes behind the roofline, for the launch shapes bench.py times alone
round=${tag%%_*}
[ "$PART" = rest ] || for spec in "S3 0 8" "S5 0 8" "S4 0 8" "S3 1 8" "S5 1 8" "S1 0 8" "S2 0 8" "S2 1 8" "S3mesh 0 8" "S5mesh 0 8" "TS 0 8" "TS 1 8"; do set -- $spec   # every line bench.py can print
  tools/pmc_roofline.sh $tag $1 $2 $3 > $out/pmc_$1_mode$2.log 2>&1 || echo "pmc $1 $2 failed"
  cp gpurun_out/${tag}_pmc_$1_mode$2.json profiles/${round}_pmc_$1_mode$2.json 2>/dev/null
  cp gpurun_out/${tag}_pmc_$1_mode$2.json $out/ 2>/dev/null
done
[ "$PART" = pmc ] && { ls $out; exit 0; }
# 2. the bench line (default invocation) and the driver's invocation
timeout -k 10 300 python bench.py > $out/bench_S3.json 2> $out/bench_S3.err || echo "bench failed"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_S3_driver.json 2>> $out/bench_S3.err
# 3. rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/prof --output-format csv -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu > $root/$out/prof.log 2>&1
cd $root
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_S3_kernel_stats.csv 2>/dev/null
rm -rf $out/prof
# 3b. grouping of short and long runs (what bench.py's default group sizes rest on)
STEPS="20 200" LANES="4" GROUPS_="1 4 10 16 20 32" bash tools/short_run_sweep.sh > /dev/null 2>&1; cp gpurun_out/short_run_sweep.log $out/short_run_sweep.log
# 4. every configuration
bash tools/run_configs.sh > $out/configs.log 2>&1
mkdir -p $out/configs; cp gpurun_out/configs/*.json $out/configs/
# 5. the multi-GPU rehearsal on one GPU
GPU_MAX_HW_QUEUES=8 timeout -k 10 400 python tools/shard_timing_direct.py > $out/shard_timing_direct.log 2>&1
GPU_MAX_HW_QUEUES=8 SHARD_PIPE_ONLY=1 SHARD_TIMING_CONFIGS="8,4,32;8,4,16;4,4,32;2,4,32;1,4,32" timeout -k 10 400 python tools/shard_timing.py > $out/shard_timing.log 2>&1
tail -12 $out/configs.log | cut -c1-330
head -4 $out/bench_S3_kernel_stats.csv
python - <<PY
import json
j=json.loads(open("$out/bench_S3.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("bench", j["ms_per_step"], j["value"], r["bound"], r["frac"], {k:v["frac"] for k,v in (r["ceilings"] or {}).items()}, r["kernel_ms"], j["latency"], j["cpu_baseline"]["value"], j["cpu_baseline"]["gpu_over_cpu"])
j=json.loads(open("$out/bench_S3_driver.json").read().strip().splitlines()[-1]); print("driver-style", j["ms_per_step"], j["value"])
PY
grep -v amdgpu.ids $out/shard_timing_direct.log | tail -8
