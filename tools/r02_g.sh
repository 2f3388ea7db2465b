#!/bin/bash
out=gpurun_out/r02g; mkdir -p $out
rm -f $out/parity.log
GLOME_PARITY_LOG=$(pwd)/$out/parity.log timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -4 $out/pytest.log
tools/pmc_roofline.sh r02g S3 0 8 > $out/pmc.log 2>&1; tail -1 $out/pmc.log | cut -c1-300
cp gpurun_out/r02g_pmc_S3_mode0.json profiles/r02_pmc_S3_mode0.json
timeout -k 10 300 python bench.py > $out/bench_S3.json 2> $out/bench_S3.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$out/bench_S3.json').read().strip().splitlines()[-1]); r=j['roofline']
print(j['ms_per_step'], j['value'], r['bound'], r['frac'], {k:v['frac'] for k,v in r['ceilings'].items()}, r['kernel_ms'], j['latency'])"
