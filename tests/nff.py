"""Test-side NFF tools: a generator of an SPD-style scene (the `balls` recursion of Eric Haines' SPD plus a polygon floor,
a patch with normals and two cones) and an independent reader that builds a SceneDesc the way the reference's Read
instances do (GlomeTrace/Data/Glome/Spd.hs:89-254), for the oracle side of the loader tests."""
import math

from glome_amd.scene import SceneDesc


def balls_nff(depth=2):
    out = ["# SPD-style test scene", "v", "from 2.125 1.25 1.75", "at 0 0 0", "up 0 0 1", "angle 45", "hither 1", "resolution 512 512",
           "b 0.078125 0.359375 0.75", "l 4 3 2", "l 1 -4 4 0.875 0.75 0.6875", "l -3 1 5",
           "f 1 0.75 0.3125 0.75 0 100 0 1", "p 4", "12 12 -0.5", "-12 12 -0.5", "-12 -12 -0.5", "12 -12 -0.5",
           "f 0.875 0.875 0.875 0.5 0.5 3.0625 0 1   # the balls"]

    def q(x):  # numbers that are exact in binary and in the decimal text alike (k / 1024), so every reader sees the same value
        return round(x * 1024) / 1024

    def rec(c, r, d, axis):
        c = tuple(q(x) for x in c); r = q(r)
        out.append("s %.10f %.10f %.10f %.10f" % (c[0], c[1], c[2], r))
        if d == 0:
            return
        for k in range(3):
            a = 2 * math.pi * k / 3 + 0.3 * d
            dirv = (math.cos(a) * 0.8, math.sin(a) * 0.8, -0.6 if axis else 0.6)
            nr = r / 3
            rec((c[0] + dirv[0] * (r + nr), c[1] + dirv[1] * (r + nr), c[2] + dirv[2] * (r + nr)), nr, d - 1, not axis)

    rec((0, 0, 0), 0.5, depth, False)
    out += ["f 0.25 0.75 0.3125 0.875 0.125 5 0.25 1.25", "c", "0.875 0.875 -0.5 0.25", "0.875 0.875 0.375 0.0625", "c", "-1.0 0.75 -0.5 0.125", "-1.0 0.75 0.625 0.125",
            "pp 3", "-0.5 -1.5 -0.375 0 0.25 1", "0.75 -1.625 -0.375 0.125 0 1", "0.125 -0.875 0.125 0 -0.25 1"]
    return "\n".join(out) + "\n"


def _tokens(text):
    for line in text.split("\n"):
        line = line.split("#", 1)[0]
        for w in line.split():
            yield w


def read_nff(text):
    """-> (SceneDesc with root/camera/lights set, background rgb).  Mirrors Spd.hs: one `tex (bih prims) fill` per fill, the
    scene = bih of those in reverse order of appearance; polygons become fans; lights default to white."""
    toks = list(_tokens(text))
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else None

    def num():
        try:
            v = float(toks[pos[0]])
        except (ValueError, IndexError):
            return None
        if not (toks[pos[0]][0].isdigit() or toks[pos[0]][0] == "-"):
            return None
        pos[0] += 1
        return v

    def vec():
        save = pos[0]
        v = [num(), num(), num()]
        if None in v:
            pos[0] = save
            return None
        return tuple(v)

    sd = SceneDesc()
    groups, lights, cam, bg = [], [], None, None
    fill, prims = None, []

    def close():
        if fill is not None:
            groups.append(sd.tex(sd.bih(list(prims)), fill))
        prims.clear()

    while peek() is not None:
        w = peek()
        if w == "v":
            pos[0] += 1
            assert toks[pos[0]] == "from"; pos[0] += 1; fr = vec()
            assert toks[pos[0]] == "at"; pos[0] += 1; at = vec()
            assert toks[pos[0]] == "up"; pos[0] += 1; up = vec()
            assert toks[pos[0]] == "angle"; pos[0] += 1; ang = num()
            assert toks[pos[0]] == "hither"; pos[0] += 2
            assert toks[pos[0]] == "resolution"; pos[0] += 3
            cam = (fr, at, up, ang)
        elif w == "l":
            pos[0] += 1
            p = vec(); c = vec() or (1.0, 1.0, 1.0)
            lights.append((p, c))
        elif w == "b":
            pos[0] += 1
            bg = vec()
        elif w == "f":
            pos[0] += 1
            c = vec(); kd, ks, shine, trans, ior = num(), num(), num(), num(), num()
            close()
            fill = sd.material_surface(c, 1 - trans, 0, kd, ks, shine)
        elif w in ("s", "c", "p", "pp") and fill is not None:
            pos[0] += 1
            if w == "s":
                c = vec(); prims.append(sd.sphere(c, num()))
            elif w == "c":
                a = vec(); ra = num(); b = vec(); rb = num()
                prims.append(sd.cone(a, ra, b, rb))
            else:
                num()
                vs, ns = [], []
                while True:
                    v = vec()
                    if v is None:
                        break
                    if w == "pp":
                        n = vec()
                        if n is None:
                            break
                        ns.append(n)
                    vs.append(v)
                fan = [sd.triangle(vs[0], vs[k], vs[k + 1]) if w == "p" else sd.trianglenorm(vs[0], vs[k], vs[k + 1], ns[0], ns[k], ns[k + 1])
                       for k in range(1, len(vs) - 1)]
                prims.append(sd.group(fan))
        else:
            break
    close()
    sd.set_root(sd.bih(groups[::-1]))
    sd.set_camera(*cam)
    for p, c in lights[::-1]:
        sd.add_light(p, c)
    return sd, bg
