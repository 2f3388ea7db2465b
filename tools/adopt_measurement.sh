#!/bin/bash
# Copy a measurement set that tools/final_measure.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call):
#   tools/adopt_measurement.sh r03_f [previous tag to remove, e.g. r03_e]
# Refuses counter passes whose clock is implausible (take those again with tools/pmc_roofline.sh TAG SCENE MODE 8).
tag=$1; prev=$2; round=${tag%%_*}
[ -d gpurun_out/$tag ] || { echo "no gpurun_out/$tag"; exit 1; }
python3 - "$tag" <<'PY' || exit 1
import glob, json, sys
bad = [f for f in glob.glob("gpurun_out/%s/%s_pmc_*.json" % (sys.argv[1], sys.argv[1])) if not (json.load(open(f)).get("clock_ghz") or 0) > 1.0]
if bad:
    print("implausible or missing clock in:", bad); sys.exit(1)
PY
for f in gpurun_out/$tag/${tag}_pmc_*.json; do b=$(basename $f); cp $f profiles/${round}_${b#${tag}_}; done
cp gpurun_out/$tag/bench_S3.json profiles/${tag}_bench_S3.json
cp gpurun_out/$tag/bench_S3_driver.json profiles/${tag}_bench_S3_driver_invocation.json
cp gpurun_out/$tag/bench_S3_kernel_stats.csv profiles/${tag}_bench_S3_kernel_stats.csv
mkdir -p profiles/${tag}_configs && cp gpurun_out/$tag/configs/*.json profiles/${tag}_configs/
grep -v amdgpu.ids gpurun_out/$tag/shard_timing.log > profiles/${tag}_shard_timing.log
grep -v amdgpu.ids gpurun_out/$tag/shard_timing_direct.log > profiles/${tag}_shard_timing_direct.log
cp gpurun_out/$tag/short_run_sweep.log profiles/${round}_short_run_sweep.log
if [ -n "$prev" ]; then rm -rf profiles/${prev}_configs profiles/${prev}_bench_S3.json profiles/${prev}_bench_S3_driver_invocation.json profiles/${prev}_bench_S3_kernel_stats.csv profiles/${prev}_shard_timing.log; fi
python3 - <<'PY'
import glob, json, sys
sys.path.insert(0, "tools")
from pmc_roofline import source_sha16
h = source_sha16(".")
for f in sorted(glob.glob("profiles/r04_pmc_*.json")):
    j = json.load(open(f)); print(f, j["clock_ghz"], "hash ok" if j["source_sha16"] == h else "STALE")
PY
