#!/bin/bash
# Bench line for every BASELINE configuration (run on the GPU box); output: gpurun_out/configs/*.json
# usage: tools/run_configs.sh [--no-cpu]
mkdir -p gpurun_out/configs
extra=$1
for spec in "S1 0" "S2 0" "S2 1" "S3 0" "S3mesh 0" "S4 0" "S3 1" "S5 0" "S5mesh 0" "S5 1" "TS 0" "TS 1"; do
  set -- $spec
  timeout -k 10 400 python bench.py --scene $1 --mode $2 $extra > gpurun_out/configs/$1_mode$2.json 2> gpurun_out/configs/$1_mode$2.err || echo "FAILED $1 $2"
  tail -c 400 gpurun_out/configs/$1_mode$2.err | grep -v amdgpu.ids | tail -2
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        c = j.get("cpu_baseline") or {}
        r = j["roofline"]
        print(f.split("/")[-1], j["value"], "Mrays/s", j["ms_per_step"], "ms", j["fps"], "fps | single frame", j["latency"]["single_frame_ms"], "ms, lone launch", j["latency"]["ms_per_frame_in_a_lone_launch"],
              "ms/frame | bound", r["bound"], r["frac"], "| rays", j["config"]["rays_per_frame"], "| cpu", c.get("value"), c.get("value_1_thread"), "x", c.get("gpu_over_cpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
