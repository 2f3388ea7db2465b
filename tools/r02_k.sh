#!/bin/bash
out=gpurun_out/r02k; mkdir -p $out
echo "== previous build"; GLOME_DEBUG_LIB=$(pwd)/tools/probe/libglome_hip_prev.so timeout -k 10 300 python tools/probe/ss_compare.py $out/prev.npz 2>&1 | grep -v amdgpu.ids
echo "== this build"; timeout -k 10 300 python tools/probe/ss_compare.py $out/new.npz 2>&1 | grep -v amdgpu.ids
python -c "
import numpy as np
a=np.load('$out/prev.npz'); b=np.load('$out/new.npz')
for k in a.files: print(k, 'SAME' if str(a[k])==str(b[k]) else 'DIFFERENT')"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adaptive or edge or s5" 2>&1 | tail -3
for s in S2 S3 S5; do timeout -k 10 300 python bench.py --scene $s --mode 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s adaptive', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'])"; done
