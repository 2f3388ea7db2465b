#!/bin/bash
out=gpurun_out/r02i; mkdir -p $out
tools/pmc_roofline.sh r02i S3 0 8 > $out/pmc.log 2>&1; tail -1 $out/pmc.log | cut -c1-200
cp gpurun_out/r02i_pmc_S3_mode0.json profiles/r02_pmc_S3_mode0.json
timeout -k 10 300 python bench.py > $out/bench_S3.json 2> $out/bench_S3.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$out/bench_S3.json').read().strip().splitlines()[-1]); r=j['roofline']
print(j['ms_per_step'], j['value'], r['bound'], r['frac'], {k:v['frac'] for k,v in r['ceilings'].items()}, r['kernel_ms'], r['wave_wait_frac'], j['latency'], j['cpu_baseline']['gpu_over_cpu'])"
for s in S1 S2 S3mesh S4 S5; do timeout -k 10 300 python bench.py --scene $s --no-cpu > $out/bench_$s.json 2>>$out/bench.err; python -c "
import json; j=json.loads(open('$out/bench_$s.json').read().strip().splitlines()[-1]); print('$s', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'], j['latency']['ms_per_frame_in_a_lone_launch'])"; done
for s in S2 S3 S5; do timeout -k 10 300 python bench.py --scene $s --mode 1 --no-cpu > $out/bench_${s}_mode1.json 2>>$out/bench.err; python -c "
import json; j=json.loads(open('$out/bench_${s}_mode1.json').read().strip().splitlines()[-1]); print('$s adaptive', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'])"; done
