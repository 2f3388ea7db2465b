#!/bin/bash
out=gpurun_out/r02n; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
for s in S4; do timeout -k 10 300 python bench.py --scene $s > $out/bench_$s.json 2>$out/bench.err; python -c "
import json,sys; j=json.loads(open('$out/bench_$s.json').read().strip().splitlines()[-1]); print('$s', j['ms_per_step'], j['value'], j['latency'], j['cpu_baseline'])"; done
