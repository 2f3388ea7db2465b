"""N2: the NFF / SPD loader (glome_sb_load_nff, a C++ restatement of Spd.hs:89-254) against an independent Python reading of
the same text fed to the oracle: same structure, same bounds, same BIH, same hits."""
import numpy as np
import pytest

import nff
from helpers import HostSim, oracle_for, random_rays
from glome_amd import api


def _load(built):
    text = nff.balls_nff(2)
    b = api.Builder()
    root, cam, lights, bg = b.load_nff(text)
    sd, bg2 = nff.read_nff(text)
    return text, b, root, cam, lights, bg, sd, bg2


def test_loader_returns_camera_lights_background(built):
    text, b, root, cam, lights, bg, sd, bg2 = _load(built)
    assert cam == ((2.125, 1.25, 1.75), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 45.0)
    assert bg == bg2 == (0.078125, 0.359375, 0.75)
    # accum_rss conses: lights come out in reverse order of appearance; a light without a colour is white
    assert lights == [((-3.0, 1.0, 5.0), (1.0, 1.0, 1.0)), ((1.0, -4.0, 4.0), (0.875, 0.75, 0.6875)), ((4.0, 3.0, 2.0), (1.0, 1.0, 1.0))]
    assert [(tuple(p), tuple(c)) for (p, c, r, s) in sd.lights] == lights


def test_loader_structure_matches_the_oracle_side_reading(built):
    text, b, root, cam, lights, bg, sd, bg2 = _load(built)
    o, om, _ = oracle_for(sd)
    # 13 spheres + 2 floor triangles + 2 cones (one cylinder-like) + 1 normal-interpolated triangle
    assert b.primcount(root) == o.primcount(om[sd.root])
    assert np.allclose(b.bound(root), o.bound(om[sd.root]), rtol=0, atol=0)
    d1, d2 = o.bih_dump(om[sd.root]), b.bih_dump(root)
    for k in range(4):  # split planes, axes, leaf sizes of the top-level bih: bit for bit
        assert np.array_equal(d1[k], d2[k])


def test_loaded_scene_traces_like_the_oracle(built):
    text, b, root, cam, lights, bg, sd, bg2 = _load(built)
    o, om, _ = oracle_for(sd)
    hs = HostSim(b, root)
    ro, rd = random_rays(400, 77, center=(0, 0, 0), radius=4, spread=1.2)
    got = hs.rayint(ro, rd)
    want = o.rayint(om[sd.root], ro.astype(np.float64), rd.astype(np.float64))
    hit = want["t"] >= 0
    assert np.array_equal(got["t"] >= 0, hit) and hit.sum() > 100
    assert np.allclose(got["t"][hit], want["t"][hit], rtol=2e-4, atol=1e-5)


def test_loader_errors_and_early_stop(built):
    b = api.Builder()
    with pytest.raises(api.GlomeError):
        b.load_nff("f 1 1 1 1 0 1 0 1\ns 0 0 0 1\n")  # no camera: readsSpdScene's pattern match fails (Spd.hs:251)
    # an unknown statement ends the parse (accum_rss falls through); what was read before it stands
    root, cam, lights, bg = b.load_nff("v\nfrom 1 1 1\nat 0 0 0\nup 0 0 1\nangle 40\nhither 1\nresolution 8 8\nb 0 0 0\nf 1 1 1 1 0 1 0 1\ns 0 0 0 1\nzzz 1 2 3\ns 5 5 5 1\n")
    assert b.primcount(root)[0] == 1 and lights == []  # (primitives, transforms, containers)
