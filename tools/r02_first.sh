#!/bin/bash
# round 2, first GPU visit: parity suite with measured levels logged, the bench line, the counter passes behind its roofline
out=gpurun_out/r02a; mkdir -p $out
rm -f $out/parity.log
GLOME_PARITY_LOG=$(pwd)/$out/parity.log timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 10 300 python tools/lone_launch.py > $out/lone_launch.log 2>&1; tail -30 $out/lone_launch.log
tools/pmc_roofline.sh r02a S3 0 8 > $out/pmc.log 2>&1; tail -3 $out/pmc.log | cut -c1-600
cp gpurun_out/r02a_pmc_S3_mode0.json profiles/r02_pmc_S3_mode0.json 2>/dev/null
timeout -k 10 300 python bench.py > $out/bench_S3.json 2> $out/bench_S3.err; echo "bench rc=$?"; tail -c 3000 $out/bench_S3.json
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_S3_20.json 2>> $out/bench_S3.err; tail -c 400 $out/bench_S3_20.json | cut -c1-400
for g in 1 2 4; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu --group $g 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps20 group', $g, j['ms_per_step'])"; done
