#!/bin/bash
# Bench line for every BASELINE configuration (run on the GPU box); output: gpurun_out/configs/*.json
mkdir -p gpurun_out/configs
for spec in "S1 0" "S2 0" "S2 1" "S3 0" "S3mesh 0" "S4 0" "S3 1"; do
  set -- $spec
  timeout -k 10 300 python bench.py --scene $1 --mode $2 --steps 400 --warmup 40 > gpurun_out/configs/$1_mode$2.json 2> gpurun_out/configs/$1_mode$2.err || echo "FAILED $1 $2"
  tail -c 400 gpurun_out/configs/$1_mode$2.err | grep -v amdgpu.ids | tail -2
done
timeout -k 10 500 python bench.py --scene S5 --mode 1 --steps 100 --warmup 10 --no-cpu > gpurun_out/configs/S5_mode1.json 2> gpurun_out/configs/S5_mode1.err || echo "FAILED S5"
timeout -k 10 300 python bench.py --scene S5 --mode 0 --steps 200 --warmup 20 --no-cpu > gpurun_out/configs/S5_mode0.json 2> gpurun_out/configs/S5_mode0.err || echo "FAILED S5 tile"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        c = j.get("cpu_baseline") or {}
        print(f.split("/")[-1], j["value"], "Mrays/s", j["ms_per_step"], "ms", j["fps"], "fps", "frac", j["roofline"]["frac"], "kernel", j["roofline"]["kernel"], j["roofline"]["kernel_ms_avg"], "rays", j["config"]["rays_per_frame"], "cpu", c.get("value"), c.get("value_1_thread"), "x", c.get("gpu_over_cpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
