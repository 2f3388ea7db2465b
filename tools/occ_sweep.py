"""bench.py (frames in flight) under the occupancy knobs GLOME_DEBUG_LB x GLOME_DEBUG_STACK_CAP.  Not a test."""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lb in ("0", "6", "8"):
    for cap in ("12", "8", "6"):
        env = dict(os.environ); env["GLOME_DEBUG_LB"] = lb; env["GLOME_DEBUG_STACK_CAP"] = cap
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu"] + sys.argv[1:], env=env, capture_output=True, text=True, cwd=HERE)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print("lb", lb, "cap", cap, "ms_per_step", j["ms_per_step"], "kernel_ms_avg", j["roofline"]["kernel_ms_avg"], flush=True)
        except Exception:
            print("lb", lb, "cap", cap, "ERR", r.stderr[-300:], flush=True)
