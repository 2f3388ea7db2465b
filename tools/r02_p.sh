#!/bin/bash
out=gpurun_out/r02p; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_gpu_c_abi" > $out/pytest1.log 2>&1; rc=$?; echo "pytest1 rc=$rc"; tail -15 $out/pytest1.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
timeout -k 10 300 python bench.py --scene S4 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S4', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'])"
