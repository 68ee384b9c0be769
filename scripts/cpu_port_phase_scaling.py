#!/usr/bin/env python3
"""Per-phase time of the GEMM-shaped CPU port at several thread counts (which pass stops scaling?).  args: [points]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bbfmm_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
pts = np.random.default_rng(42).random((n, 3)); w = np.random.default_rng(43).random((n, 1))
tree = O.FmmTree(pts, 7, 0, True, True); tree.gemm_shaped = True
hw = int(O.lib().oracle_num_threads())
out = {"points": n, "hardware_threads": hw, "phases_s": {}}
wf = tree._w(w)
for th in sorted({hw, hw // 2, hw // 4, hw // 8}, reverse=True):
    O.lib().oracle_set_num_threads(th)
    tree.set_weights(w); tree.evaluate(w, pts)
    rec = {}
    t0 = time.time(); tree.set_weights(w); rec["set_weights(P2M+M2M)"] = time.time() - t0
    t0 = time.time(); tp, tgt_ptr, tgt_idx, cell_of = tree._assign_targets(pts); rec["assign_targets(python)"] = time.time() - t0
    t0 = time.time(); active = tree._ancestor_flags(np.unique(cell_of)); rec["ancestor_flags(python)"] = time.time() - t0
    t0 = time.time(); tree._downward(wf, active); rec["downward(M2L+P2L+L2L)"] = time.time() - t0
    for name, fl in (("P2P", 1), ("M2P", 2), ("L2P", 4)):
        t0 = time.time(); tree._leaf_pass(wf, tp, tgt_ptr, tgt_idx, False, fl); rec["leaf_" + name] = time.time() - t0
    out["phases_s"][str(th)] = {k: round(v, 3) for k, v in rec.items()}
print(json.dumps(out))
