#!/bin/bash
# GPU box: the parity suite, then GlomeView's default scene in both render modes (a quick A/B for generic-tier changes)
tag=${1:-ts}
PYTHONUNBUFFERED=1 timeout -k 10 600 python -m pytest tests -m gpu -x -v > gpurun_out/${tag}_pytest.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest.log
for mode in 0 1; do
  timeout -k 10 200 python bench.py --scene TS --mode $mode --no-cpu > gpurun_out/${tag}_TS$mode.json 2> gpurun_out/${tag}_TS$mode.err || exit 1
done
python - <<PY
import json
for mode in (0, 1):
    j = json.loads(open("gpurun_out/${tag}_TS%d.json" % mode).read().strip().splitlines()[-1])
    print("TS mode", mode, j["ms_per_step"], "ms", j["value"], "Mrays/s, single frame", j["latency"]["single_frame_ms"])
PY
