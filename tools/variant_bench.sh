#!/bin/bash
# A/B/C.. of library variants on the GPU box (tools/build_variants.py): each variant and the in-tree build ("base"), alternating,
# two rounds.   usage: tools/variant_bench.sh "v1 v2 base" "S3 0" "S5 0" ...
variants=$1; shift
for spec in "$@"; do set -- $spec
  for round in 1 2; do for v in $variants; do
    if [ $v = base ]; then unset GLOME_DEBUG_LIB; else export GLOME_DEBUG_LIB=glome_amd/variants/$v.so; fi
    timeout -k 10 300 python bench.py --scene $1 --mode $2 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $1 mode $2', j['ms_per_step'], j['value'], 'single', j['latency']['single_frame_ms'], 'lone', j['latency']['ms_per_frame_in_a_lone_launch'], flush=True)" || exit 1
  done; done
done
