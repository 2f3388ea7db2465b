#!/bin/bash
# Where round 4's first measurement set lost 10-13 % on the Mesh walks against round 3's: the gated builds of tools/build_variants.py
# (and round 3's own library) side by side, alternating, in one gpurun call.    usage: tools/probe/mesh_regress_ab.sh "S5mesh 0" ...
for spec in "$@"; do set -- $spec
  for round in 1 2; do for which in r03 base nointerleave rb_no_kernarg rb_no_repixel rb_old_take rb_old_count; do
    unset GLOME_DEBUG_LIB GLOME_DEBUG_NO_INTERLEAVE
    case $which in base) ;; nointerleave) export GLOME_DEBUG_NO_INTERLEAVE=1 ;; *) export GLOME_DEBUG_LIB=glome_amd/variants/$which.so ;; esac
    timeout -k 10 300 python bench.py --scene $1 --mode $2 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $1 mode $2', j['ms_per_step'], j['value'], 'single', j['latency']['single_frame_ms'], 'lone', j['latency']['ms_per_frame_in_a_lone_launch'], flush=True)" || echo "$which failed"
  done; done
done
