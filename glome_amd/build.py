"""Build libglome_hip.so in-tree: the host half with g++, the HIP half with hipcc for gfx950 only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libglome_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=on: contraction decided per source expression, so every kernel instance rounds identically (the tests
# require bit-identical frames across instances); denormals flushed so 1/x is a bare v_rcp_f32
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on",
            "-fgpu-flush-denormals-to-zero"] + os.environ.get("GLOME_EXTRA_HIPFLAGS", "").split()


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(HERE, "..", "include", "glome_hip.h"))
    host_src = os.path.join(CSRC, "capi_host.cpp")
    dev_src = os.path.join(CSRC, "glome_device.hip")
    host_o = os.path.join(CSRC, "capi_host.o")
    dev_o = os.path.join(CSRC, "glome_device.o")

    def run(cmd):
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if force or _stale(host_o, [host_src] + hdrs):
        run(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", host_src, "-o", host_o])
    if force or _stale(dev_o, [dev_src] + hdrs):
        run([HIPCC] + HIPFLAGS + ["-c", dev_src, "-o", dev_o])
    if force or _stale(LIB, [host_o, dev_o]):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, host_o, dev_o])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
