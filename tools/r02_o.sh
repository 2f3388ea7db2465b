#!/bin/bash
for lb in 1 2 3 4; do GLOME_DEBUG_LB=$lb timeout -k 10 300 python bench.py --scene S4 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S4 LB=$lb', j['ms_per_step'], j['value'], j['latency']['single_frame_ms'])"; done
