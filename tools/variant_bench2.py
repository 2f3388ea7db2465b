"""bench.py (frames in flight) once per library variant; prints ms_per_step.  Not a test."""
import glob, json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(HERE, "glome_amd", "variants", "libglome_*.so"))):
    env = dict(os.environ); env["GLOME_DEBUG_LIB"] = lib
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu"] + sys.argv[1:], env=env, capture_output=True, text=True, cwd=HERE)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(os.path.basename(lib), "ms_per_step", j["ms_per_step"], "value", j["value"], "kernel_ms_avg", j["roofline"]["kernel_ms_avg"], flush=True)
    except Exception as e:
        print(os.path.basename(lib), "ERR", r.stderr[-300:], flush=True)
