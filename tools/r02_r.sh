#!/bin/bash
out=gpurun_out/r02r; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $out/pytest.log
for s in S3 S4; do timeout -k 10 300 python bench.py --scene $s --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'])"; done
