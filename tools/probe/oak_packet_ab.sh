P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["value"], "single", j["latency"]["single_frame_ms"], "lone", j["latency"]["ms_per_frame_in_a_lone_launch"], flush=True)'
for round in 1 2; do for m in 0 1; do
  GLOME_DEBUG_NO_ITEM_PACKETS=1 timeout -k 10 300 python bench.py --scene TS --mode $m --no-cpu 2>/dev/null | python -c "$P" "per-lane oak, mode $m"
  timeout -k 10 300 python bench.py --scene TS --mode $m --no-cpu 2>/dev/null | python -c "$P" "oak as packet, mode $m"
done; done
