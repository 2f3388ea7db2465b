"""Static look at what the compiler made of each kernel of one part of glome_device.hip: registers, spills, scratch bytes
and the count of every kind of memory instruction (flat_load where global_load / s_load was meant is a pool base the
compiler could not place: DESIGN.md 4.4a).

  python tools/kernel_mix.py PART [-DFLAG ...]        (runs here: hipcc cross-compiles; output under /tmp/kernel_mix/pPART)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glome_amd import build  # noqa: E402

KINDS = ("flat_load", "flat_store", "global_load", "global_store", "scratch_load", "scratch_store", "s_load", "s_buffer_load", "ds_", "v_readlane", "v_writelane", "s_waitcnt")


def main():
    part = int(sys.argv[1])
    out = f"/tmp/kernel_mix/p{part}"
    os.makedirs(out, exist_ok=True)
    cmd = [build.HIPCC] + build.HIPFLAGS + [f"-DGLOME_PART={part}", "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "glome_amd/csrc/glome_device.hip"), "-o", "p.o",
                                            "-Rpass-analysis=kernel-resource-usage", "-save-temps"] + sys.argv[2:]
    r = subprocess.run(cmd, cwd=out, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(1)
    usage, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2); usage[cur] = {}
        elif cur:
            usage[cur][m.group(1).split(" [")[0]] = m.group(2)
    asm = open(os.path.join(out, "glome_device-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    body, name = {}, None
    for line in asm:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); body[name] = []
        elif name:
            body[name].append(line.strip())
            if line.strip().startswith("s_endpgm") or line.strip().startswith(".end_amdhsa_kernel"):
                name = None
    for k, u in usage.items():
        lines = body.get(k, [])
        demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        mix = {kind: sum(1 for x in lines if x.startswith(kind)) for kind in KINDS}
        print(demangled)
        print("   ", " ".join(f"{a}={b}" for a, b in u.items()), f"instructions={sum(1 for x in lines if x and not x.startswith((';', '.')) and not x.endswith(':'))}")
        print("   ", " ".join(f"{a}={b}" for a, b in mix.items() if b))


if __name__ == "__main__":
    main()
