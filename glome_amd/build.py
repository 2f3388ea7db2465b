"""Build libglome_hip.so in-tree: the host half with g++, the HIP half with hipcc for gfx950 only.

glome_device.hip is compiled once per PART (-DGLOME_PART=k, see the top of that file), the parts in parallel: the kernel
instances are what takes the time (one translation unit: 4.5 minutes; twelve parts on 8 cores: about one and a half)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libglome_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
NPARTS = 12  # == kParts in glome_device.hip
# -ffp-contract=on: contraction decided per source expression, so every kernel instance rounds identically (the tests
# require bit-identical frames across instances); denormals flushed so 1/x is a bare v_rcp_f32
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on",
            "-fgpu-flush-denormals-to-zero"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True, lib=LIB, extra_flags=None, obj_dir=None, jobs=None):
    """lib / extra_flags / obj_dir: a variant build beside the in-tree one (A/B measurements through GLOME_DEBUG_LIB)."""
    extra = list(extra_flags) if extra_flags is not None else os.environ.get("GLOME_EXTRA_HIPFLAGS", "").split()
    # a build with extra flags is a VARIANT: its objects and its library live beside the in-tree ones, named by the flags (a stable
    # hash: the same flags find their objects again), so the in-tree library never silently holds variant objects
    tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] if extra else ""
    obj_dir = obj_dir or (OBJ if not extra else OBJ + "_" + tag)
    if extra and lib == LIB:
        os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
        lib = os.path.join(HERE, "variants", "flags_" + tag + ".so")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(HERE, "..", "include", "glome_hip.h"))
    host_src = os.path.join(CSRC, "capi_host.cpp")
    dev_src = os.path.join(CSRC, "glome_device.hip")
    host_o = os.path.join(obj_dir, "capi_host.o")
    part_o = [os.path.join(obj_dir, "glome_device_p%d.o" % k) for k in range(NPARTS)]

    def run(cmd):
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    todo = []
    if force or _stale(host_o, [host_src] + hdrs):
        todo.append(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall"] + [f for f in extra if f.startswith("-D")] + ["-c", host_src, "-o", host_o])
    for k, o in enumerate(part_o):
        if force or _stale(o, [dev_src] + hdrs):
            todo.append([HIPCC] + HIPFLAGS + extra + ["-DGLOME_PART=%d" % k, "-c", dev_src, "-o", o])
    if todo:
        jobs = jobs or int(os.environ.get("GLOME_BUILD_JOBS", "0")) or min(len(todo), os.cpu_count() or 1)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(run, todo))
    if force or _stale(lib, [host_o] + part_o):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, host_o] + part_o)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
