"""An independent reading of the text GlomeView prints for a scene (`show geom`, Glome.hs:431) -- test infrastructure.

hs_double: GHC's `show :: Double -> String` (GHC.Float.showFloat) restated over Python's shortest-digits repr.
parse: a table-driven reader of Haskell `Show` output into nested tuples.  A SolidItem shows as "SI " ++ show s without
parentheses wherever it stands (Solid.hs:277-278), so application cannot be delimited by parentheses alone: the reader
knows each constructor's arity (the data declarations cited in include/glome_hip.h)."""
import decimal
import math
import re

ARITY = {"SI": 1, "Sphere": 3, "Triangle": 3, "TriangleNorm": 6, "Box": 1, "Plane": 2, "Disc": 3, "Cylinder": 3, "Cone": 4, "Instance": 2,
         "Difference": 3, "Intersection": 1, "Bound": 2, "InnerBound": 2, "Tex": 2, "NoShadow": 1, "OnlyShadow": 1, "Void": 0, "Vec": 3,
         "Matrix": 12, "Xfm": 2, "BihLeaf": 1, "BihBranch": 5, "Leaf": 1, "Branch": 4, "Tri": 8, "Texture": 0, "True": 0, "False": 0, "Mesh": 4,
         "Infinity": 0, "NaN": 0}
RECORDS = {"Bbox": ("p1", "p2"), "Bih": ("bihbb", "bihroot")}
TOKEN = re.compile(r"\s*(-?\d+\.\d+(?:e-?\d+)?|-?\d+|-?Infinity|[A-Za-z_][A-Za-z0-9_']*|[()\[\]{},=<>])")


def hs_double(x):
    """show (x :: Double): shortest digits ds and exponent e with x = 0.ds * 10^e; fixed notation iff 0 <= e <= 7."""
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "-Infinity" if x < 0 else "Infinity"
    sign = "-" if math.copysign(1.0, x) < 0 else ""
    x = abs(x)
    if x == 0:
        return sign + "0.0"
    _, dg, exp = decimal.Decimal(repr(x)).as_tuple()  # repr: the shortest digits that read back to x, like floatToDigits
    digits = "".join(map(str, dg)).lstrip("0")
    e = len(digits) + exp
    digits = digits.rstrip("0") or "0"
    if e < 0 or e > 7:
        return sign + digits[0] + "." + (digits[1:] or "0") + "e" + str(e - 1)
    if e == 0:
        return sign + "0." + digits
    whole = digits[:e].ljust(e, "0")
    return sign + whole + "." + (digits[e:] or "0")


def tokens(text):
    pos, out = 0, []
    while pos < len(text):
        if text[pos:].strip() == "":
            break
        m = TOKEN.match(text, pos)
        if not m:
            raise ValueError("bad text at %d: %r" % (pos, text[pos:pos + 30]))
        out.append(m.group(1))
        pos = m.end()
    return out


class _P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def take(self, want=None):
        tok = self.t[self.i]
        if want is not None and tok != want:
            raise ValueError("expected %r, got %r at token %d" % (want, tok, self.i))
        self.i += 1
        return tok

    def value(self):
        tok = self.peek()
        if tok == "(":
            self.take()
            v = self.value()
            self.take(")")
            return v
        if tok == "[":
            self.take()
            items = []
            if self.peek() != "]":
                items.append(self.value())
                while self.peek() == ",":
                    self.take()
                    items.append(self.value())
            self.take("]")
            return items
        if tok == "<":
            self.take(); self.take("Tag")
            v = self.value()
            self.take(">")
            return ("Tag", v)
        self.take()
        if re.fullmatch(r"-?\d+", tok):
            return int(tok)
        if tok[0].isdigit() or tok[0] == "-":
            return float(tok.replace("Infinity", "inf"))
        if tok in RECORDS:
            self.take("{")
            vals = []
            for k, field in enumerate(RECORDS[tok]):
                if k:
                    self.take(",")
                self.take(field); self.take("=")
                vals.append(self.value())
            self.take("}")
            return (tok,) + tuple(vals)
        if tok == "Infinity":
            return math.inf
        if tok == "NaN":
            return math.nan
        if tok == "Mesh":  # Mesh.hs:44-46: four fields shown at precedence 0, so the bbox and the BVH stand bare
            return ("Mesh", self.value(), self.value(), self.value(), self.value())
        if tok not in ARITY:
            raise ValueError("unknown constructor %r" % tok)
        return (tok,) + tuple(self.value() for _ in range(ARITY[tok])) if ARITY[tok] else tok


def parse(text):
    p = _P(tokens(text))
    v = p.value()
    if p.i != len(p.t):
        raise ValueError("text after the value")
    return v


def walk(v):
    """Every constructor application of a parsed value, preorder."""
    if isinstance(v, tuple):
        yield v
        for a in v[1:]:
            yield from walk(a)
    elif isinstance(v, list):
        for a in v:
            yield from walk(a)
