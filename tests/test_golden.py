"""Committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py from the fp64 oracle).
CPU suite: the oracle still reproduces them exactly, and the host-compiled device code matches them within the fp32
tolerance.  The GPU suite (test_gpu_parity.py) checks the C ABI against the same files."""
import glob
import json
import os

import numpy as np
import pytest

from helpers import HostSim, oracle_for, product_camera_lights
from glome_amd import api

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "*.json")))


def load_gold(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg, json.load(open(os.path.join(GOLD, name + ".json")))


def compare_backend_to_gold(g, got, got_shadow, img, nm, same_prim_min=0.97, normal_atol=2e-3):
    t = np.asarray(g["t"])
    assert np.array_equal(got["t"] >= 0, t >= 0)
    h = t >= 0
    assert np.allclose(got["t"][h], t[h], rtol=2e-4, atol=1e-5)
    inv = np.full(max(nm) + 2, -1); inv[np.asarray(nm)] = np.arange(len(nm))
    gp = np.asarray(g["prim"])[h]
    same = (inv[got["prim"][h]] == gp) & (gp >= 0)  # (primitives a constructor made inside the backend -- flatten_transform's -- have no common id)
    assert same[gp >= 0].mean() > same_prim_min  # (a scene may state a lower bar with its reason: parity.same_prim_min)
    assert np.array_equal(got["tex"][h][same], np.asarray(g["tex"])[h][same])
    assert np.allclose(got["n"][h][same], np.asarray(g["n"])[h][same], atol=normal_atol)
    assert np.array_equal(got_shadow, np.asarray(g["shadow"], bool))
    ref = np.asarray(g["image"]["rgbad"]).reshape(g["image"]["h"], g["image"]["w"], 5)
    e = (np.abs(img[..., :4] - ref[..., :4]) / np.maximum(1, np.abs(ref[..., :4]))).max(-1)
    assert np.mean(e > 1e-4) <= 0.03 and np.median(e) < 1e-5


def test_fixtures_exist():
    assert len(NAMES) >= 10


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(built, name):
    mg, g = load_gold(name)
    fresh = mg.make(name)
    for k in ("t", "prim", "n", "tex", "shadow"):
        assert fresh[k] == g[k], k
    assert fresh["image"]["rgbad"] == g["image"]["rgbad"] and fresh["image"]["packed"] == g["image"]["packed"] and fresh["image"]["rays"] == g["image"]["rays"]


@pytest.mark.parametrize("name", NAMES)
def test_device_code_on_host_matches_golden(built, name):
    mg, g = load_gold(name)
    sd = mg.SCENES[name]()
    b = api.Builder()
    nm, _ = sd.replay(b)
    hs = HostSim(b, nm[sd.root])
    ro, rd = mg.golden_inputs()
    cam, lights = product_camera_lights(sd)
    img, _ = hs.render(cam, lights, g["image"]["w"], g["image"]["h"], g["image"]["maxdepth"])
    compare_backend_to_gold(g, hs.rayint(ro, rd), hs.shadow(ro, rd, np.full(len(ro), g["shadow_tmax"], np.float32)), img, nm, getattr(sd, "same_prim_min", 0.97), getattr(sd, "normal_atol", 2e-3))
