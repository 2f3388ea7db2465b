#!/bin/bash
# what each item of GlomeView's default scene costs: the frame period without it (bench.py --scene TSparts, GLOME_TS_SKIP)
for skip in "" 0 1 2 3 4 5 6 7 "4,5" "0,1,2,3,6,7"; do
  GLOME_TS_SKIP=$skip timeout -k 10 280 python bench.py --scene TSparts --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('without [$skip]', j['ms_per_step'], 'ms  single', j['latency']['single_frame_ms'], j['config']['rays_per_frame'])"
done
