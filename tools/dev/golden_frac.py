"""GPU probe: the fraction of pixels of every golden path-traced image that the HIP kernels decide differently (what
tests/test_pt_golden.py bounds), printed — to compare builds.  usage: golden_frac.py"""
import os
import sys

R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from __graft_entry__ import load_package
import test_pt_golden as T

pkg = load_package()
for name in T.GOLDEN_IMAGES:
    g, first, img, counts = T.render_golden(pkg.RenderContext(device=0), pkg, name)
    rel = 5e-3 if "cards" in name else 0.0
    d0 = (np.abs(first - g["sample0"]) - rel * np.abs(g["sample0"])).max(-1)
    d = (np.abs(img - g["image"]) - rel * np.abs(g["image"])).max(-1)
    print("%-28s sample 0: %3d pixels differ   4 spp: %3d of %d" % (name, int((d0 > 1e-3).sum()), int((d > 1e-3).sum()), d.size), flush=True)
