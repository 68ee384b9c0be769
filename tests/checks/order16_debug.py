"""Order-16 M2L debugging aid: L of the device against the oracle per class of level 2, under the given switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ferreus_rbf_rs_amd as F
from conftest import inject_product_operators, relerr
from oracle import bbfmm_oracle as O
O.build_passes()
order = int(sys.argv[1])
pts = np.random.default_rng(300 + order).random((2200, 3))
params = (120, 2, 1e-9, 1024)
t = F.FmmTree(pts, order, F.KernelParams(F.KernelType(0)), True, True, params=F.FmmParams(*params))
r = O.FmmTree(pts, order, 0, True, True, None, O.FmmParams(*params))
inject_product_operators(t, r)
w = np.random.default_rng(0).random((2200, 1))
t.set_weights(w); r.set_weights(w)
print("M", relerr(t.debug_get_coefficients("M", 1), r.M))
y, yr = t.evaluate(w, pts), r.evaluate(w, pts)
L = t.debug_get_coefficients("L", 1)[0]
Lr = r.L[0]
print("L", relerr(L, Lr), "y", relerr(y, yr), "ranks", t.m2l_ranks()[2])
n = L.shape[1]
for c in range(9, 73, 9):
    d = np.abs(L[c] - Lr[c]); ratio = L[c] / np.where(Lr[c] == 0, 1, Lr[c])
    print(c, "max rel", d.max() / np.abs(Lr[c]).max(), "ratio first/mid/last", ratio[0], ratio[n // 2], ratio[-1],
          "err by node block of 512:", [float("%.2g" % (d[i:i + 512].max() / np.abs(Lr[c]).max())) for i in range(0, n, 512)])
