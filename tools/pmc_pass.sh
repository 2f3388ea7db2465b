#!/bin/bash
# usage: tools/pmc_pass.sh OUTDIR COUNTERSET...   (each COUNTERSET = space separated counters for one rocprofv3 --pmc pass)
# Runs tools/pmc_run.py under rocprofv3 once per counter set (kernel-trace only, as the pool requires).
out=$1; shift
root=$(pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d "$root/$out/pass$i" --output-format csv -- python3 "$root/tools/pmc_run.py" > "$root/$out/pass$i.log" 2>&1 || exit 1
done
