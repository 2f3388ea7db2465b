"""Counters behind bench.py's per-ceiling roofline: summarise rocprofv3 --pmc passes (tools/pmc_roofline.sh) of the launch
shape bench.py times alone -- one launch in flight, GROUP frames per launch -- into profiles/<tag>_pmc_<scene>_mode<m>.json.

  python tools/pmc_roofline.py PASSDIR SCENE MODE GROUP OUT.json

Per counter: the mean over the dispatches of the dominant kernel (the kernel with the largest total duration).  FETCH_SIZE / WRITE_SIZE stay in KiB as rocprofv3
reports them (bench.py applies the gfx950 correction: FETCH_SIZE x 2, MI355X_MICROARCH.md HBM section).  clock_ghz =
GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (same guide, DVFS paragraph), from the pass that carries GRBM_GUI_ACTIVE."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def source_sha16(root):
    """what bench.py compares a profile with: a hash of the kernel sources the library is built from"""
    h = hashlib.sha256()
    for d in ("glome_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(root, d))):
            if f.endswith((".hpp", ".h", ".hip", ".cpp")):
                h.update(f.encode()); h.update(open(os.path.join(root, d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    passdir, scene, mode, group, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rows = []
    for f in sorted(glob.glob(passdir + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            rows.append((f, r))
    # dominant kernel family: by total duration over distinct dispatches
    dur = collections.defaultdict(float)
    seen = set()
    for f, r in rows:
        key = (f, r["Dispatch_Id"])
        if key in seen:
            continue
        seen.add(key)
        dur[r["Kernel_Name"].split("(")[0]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    fam = max((k for k in dur if ("k_ss_frame" in k if mode != 0 else "k_render" in k)), key=lambda k: dur[k])
    per_pass = collections.defaultdict(lambda: collections.defaultdict(list))  # file -> counter -> values per dispatch
    times = collections.defaultdict(dict)
    names = set()
    for f, r in rows:
        kn = r["Kernel_Name"].split("(")[0]
        if fam not in kn:
            continue
        names.add(kn)
        per_pass[f][r["Counter_Name"]].append(float(r["Counter_Value"]))
        times[f][r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6  # ms
    counters, kernel_ms, clock = {}, [], None
    per_unit = 1  # dispatches that make one frame group (the adaptive sampler, too, is one launch per frame)
    for f, cs in per_pass.items():
        t = list(times[f].values())
        n_units = max(1, len(t) // per_unit)
        ms = sum(t) / n_units
        kernel_ms.append(ms)
        for c, v in cs.items():
            counters[c] = sum(v) / n_units
        if "GRBM_GUI_ACTIVE" in cs:
            clock = (sum(cs["GRBM_GUI_ACTIVE"]) / n_units) / 8.0 / (ms * 1e-3) / 1e9
    note = None
    if clock is not None and not (1.0 < clock < 2.6):  # (the GPU's top clock is 2.4 GHz: a pass that reads more counted something else as well)
        note = "GRBM_GUI_ACTIVE of this pass gives %.2f GHz: discarded, bench.py uses its default clock" % clock
        clock = None
    res = {"scene": scene, "mode": mode, "kernel": sorted(names), "frames_per_launch": group, "launches_per_pass": max(len(t) // per_unit for t in times.values()) if times else 0,
           "kernel_ms": round(sum(kernel_ms) / max(1, len(kernel_ms)), 4), "kernel_ms_per_pass": [round(x, 4) for x in kernel_ms], "clock_ghz": round(clock, 4) if clock else None, **({"clock_note": note} if note else {}),
           "counters": {k: round(v, 1) for k, v in sorted(counters.items())},
           "source_sha16": source_sha16(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
           "how": "tools/pmc_roofline.sh: one rocprofv3 --kernel-trace --pmc pass per counter group over tools/pmc_run.py (lanes 1: one launch in flight); means per launch of the dominant kernel"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
