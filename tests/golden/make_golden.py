#!/usr/bin/env python3
"""Generate tests/golden/*.json from the CPU oracle (fp64).

The reference (Haskell) ships no golden vectors and cannot be run in this image ("parity unpinned"), so
these fixtures are outputs of the oracle itself, frozen so that (a) a later change to the oracle that alters any
reference semantics shows up as a diff, and (b) the HIP path is compared against committed numbers as well as against
the live oracle.  Inputs are regenerated from seeds; only expected outputs are stored.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import zoo  # noqa: E402
from helpers import oracle_for, random_rays  # noqa: E402
from glome_amd import scenes  # noqa: E402

N_RAYS, SEED, W, H, MAXDEPTH = 96, 2024, 32, 18, 3
SCENES = dict(zoo.ALL)
SCENES.update({"S1": lambda: scenes.s1(nlights=2), "S3small": lambda: scenes.s3(24), "S3mesh_small": lambda: scenes.s3(24, as_mesh=True), "S4": scenes.s4})


def golden_inputs():
    return random_rays(N_RAYS, SEED, center=(0, 1.5, 0), radius=13, spread=7)


def make(name):
    sd = SCENES[name]()
    o, om, _ = oracle_for(sd)
    ro, rd = golden_inputs()
    r = o.rayint(om[sd.root], ro.astype(np.float64), rd.astype(np.float64))
    inv = np.full(max(om) + 2, -1); inv[np.asarray(om)] = np.arange(len(om))
    tm = np.full(N_RAYS, 12.0)
    sh = o.shadow(om[sd.root], ro.astype(np.float64), rd.astype(np.float64), tm)
    img, packed, cnt = o.render(W, H, maxdepth=MAXDEPTH)
    return {
        "scene": name, "n_rays": N_RAYS, "seed": SEED, "shadow_tmax": 12.0,
        "t": [float(x) for x in r["t"]],
        "prim": [int(inv[p]) if p >= 0 else -1 for p in r["prim"]],  # SceneDesc-local ids (backend independent)
        "n": [[float(x) for x in v] for v in r["n"]],
        "tex": [[int(x) for x in v] for v in r["tex"]],
        "shadow": [int(x) for x in sh],
        "image": {"w": W, "h": H, "maxdepth": MAXDEPTH, "rgbad": [float(x) for x in img.ravel()], "packed": [int(x) for x in packed.ravel()],
                  "rays": [cnt["rays_primary"], cnt["rays_shadow"], cnt["rays_secondary"]], "bih_nodes": cnt["bih_nodes"], "prim_tests": cnt["prim_tests"]},
    }


if __name__ == "__main__":
    for name in sorted(SCENES):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(make(name), f)
        print("wrote", name)
