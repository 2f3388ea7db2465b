#!/bin/bash
# usage: tools/probe/pmc_icache.sh SCENE MODE   -- instruction-cache counters of the render kernel (one rocprofv3 --pmc pass)
scene=${1:-TS}; mode=${2:-0}
root=$(pwd); out=$root/gpurun_out/icache_${scene}_$mode; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SCENE=$scene MODE=$mode GROUP=8 LAUNCHES=6 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out --output-format csv -- python3 $root/tools/pmc_run.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
cd $root
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if "render" in k or "ss_frame" in k: print(k, {a: "%.4g" % b for a, b in c.items()})
PY
