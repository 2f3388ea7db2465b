#!/bin/bash
out=gpurun_out/r02h; mkdir -p $out
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "soup or two_row or full_size_faithful or S3small" > $out/pytest1.log 2>&1; rc=$?; echo "pytest1 rc=$rc" | tee -a $out/pytest1.log; tail -5 $out/pytest1.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/pytest.log; tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit 1
NF=1,8 PERCU=0,24 timeout -k 10 120 python tools/lone_launch.py 2>/dev/null | tee -a $out/lone.log
timeout -k 10 120 python bench.py --no-cpu > $out/bench.json 2>$out/bench.err; python -c "
import json; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['latency'])"
