#!/bin/bash
out=gpurun_out/r02m; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "materials or textures or golden or soup" > $out/pytest1.log 2>&1; rc=$?; echo "pytest1 rc=$rc"; tail -4 $out/pytest1.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
for s in S3 S4; do timeout -k 10 300 python bench.py --scene $s --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'], j['latency']['ms_per_frame_in_a_lone_launch'])"; done
