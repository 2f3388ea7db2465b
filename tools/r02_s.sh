#!/bin/bash
out=gpurun_out/r02s; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "soup or two_row or full_size_faithful or S3small or mirror or s5_4k_tile" > $out/pytest1.log 2>&1; rc=$?; echo "pytest1 rc=$rc"; tail -3 $out/pytest1.log
[ $rc -ne 0 ] && exit 1
for s in S3 S5; do timeout -k 10 300 python bench.py --scene $s --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'], j['latency']['ms_per_frame_in_a_lone_launch'])"; done
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', j['ms_per_step'], j['value'])"
