#!/usr/bin/env python3
"""Build named variants of the library beside the in-tree one (glome_amd/variants/<name>.so, git-ignored; they travel to the GPU
box) for A/B measurements through GLOME_DEBUG_LIB.   usage: tools/build_variants.py name="-DFLAG=1 -DOTHER=2" ..."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from glome_amd import build as B  # noqa: E402

out = os.path.join(HERE, "..", "glome_amd", "variants")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    lib = os.path.join(out, name + ".so")
    B.build(lib=lib, extra_flags=flags.split(), obj_dir=os.path.join(out, "obj_" + name), verbose=False)
    print("built", lib, flags, flush=True)
