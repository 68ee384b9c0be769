#!/usr/bin/env python3
"""Device vs oracle (same operators) on the soak run's 12-cluster cloud, L coefficients level by level: regular / adaptive
tree, ACA / uncompressed, orders 5 and 7 (sparse coarse levels, classes with a handful of cells)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O
from conftest import inject_product_operators, relerr
def clustered(rng, n, d):
    k = 12
    c = rng.random((k, d)); s = 0.01 + 0.08 * rng.random(k); which = rng.integers(0, k, n)
    return np.clip(c[which] + rng.normal(size=(n, d)) * s[which, None], 0.0, 0.999)
n = 100000
rng = np.random.default_rng(123)
pts = clustered(rng, n, 3)
w = rng.random((n, 1))
for adaptive, order, comp in [(False, 7, 2), (True, 7, 0), (True, 5, 2), (True, 7, 2)]:
    prm = (256, comp, 10.0 ** -order, 1024)
    t = F.FmmTree(pts, order, F.KernelParams(F.KernelType(0)), adaptive, True, params=F.FmmParams(*prm))
    r = O.FmmTree(pts, order, 0, adaptive, True, None, O.FmmParams(*prm))
    inject_product_operators(t, r)
    t.set_weights(w); r.set_weights(w)
    y = t.evaluate(w, pts); yr = r.evaluate(w, pts)
    keys, leaf = t.cells(); lev = (keys & 0x7FFF).astype(int)
    Ld = t.debug_get_coefficients("L", 1)[0]; Lo = r.L[0]
    per = {}
    for l in range(2, t.stats().depth + 1):
        m = lev == l
        if m.any(): per[l] = float(np.abs(Ld[m] - Lo[m]).max() / np.abs(Lo[m]).max())
    st = t.stats()
    print(json.dumps({"adaptive": adaptive, "order": order, "compression": comp, "depth": st.depth, "n_v": st.n_v, "n_x": st.n_x, "L_err_by_level": per, "y_err": relerr(y, yr)}))
