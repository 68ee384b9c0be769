#!/usr/bin/env python3
"""Round-6 fuzz on the GPU box: random SEQUENCES of the evaluator's calls on a (plain) handle against the same sequence on the
oracle's restatement of the reference's FmmTree (oracle/bbfmm_oracle.py: bbfmm.rs:383-616, rbf.rs:1338-1379).  The single-call
parity tests fix the state a call starts from; here the state is whatever the calls before left behind -- the multipoles of
the last set_weights under another call's weights (the reference's mixture: old far field, new near field), a matvec between
set_local_coefficients and evaluate_leaves, row subsets, right-hand-side counts that change, targets outside the tree.  After
every call: the same values at 1e-11 of the call's largest value (the oracle runs the product's own host-computed M2L factors, so
only summation order differs), or the same refusal with the same offending row.

Calls whose outcome the reference leaves open are made on both sides and not compared: leaves-only calls without a
set_local_coefficients since the last call that rewrote the expansions (the header's contract), and calls whose weights have
another column count than set_weights' (the reference indexes past its arrays).

  python tests/checks/handle_sequence_fuzz.py [sequences] [seed] [calls per sequence]   -> JSON lines, last line = summary"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O
from conftest import inject_product_operators

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(seed)
TOL = 1e-11
O.build_passes()


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


def run(f):
    try:
        return f(), None
    except (ValueError, O.PointOutsideTree, O.KernelDoesNotSupportGradients) as e:
        return None, (getattr(e, "point_index", None), type(e).__name__, str(e)[:100])


fails = 0
for s in range(n_seq):
    d = [3, 3, 3, 2, 2, 1][int(rng.integers(0, 6))]
    n = int(rng.integers(3000, 12000)) if d > 1 else int(rng.integers(500, 3000))
    # the solver's tree (adaptive, sparse, extents from the data: rbf.rs:459-467) most of the time; the evaluator's otherwise
    # (rbf.rs:594-631: explicit extents, not sparse) and regular trees
    adaptive, sparse = bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 3) > 0)
    extents = None if rng.integers(0, 3) else [-0.25] * d + [1.5] * d
    pts = np.vstack([rng.random((n // 2, d)), np.clip(rng.normal(size=(n - n // 2, d)) * 0.06 + 0.4, 0.0, 0.999)])
    pts = np.unique(pts, axis=0)
    n = len(pts)
    kid = [0, 1, 2, 3, 7][int(rng.integers(0, 5))]            # 7: a kernel without gradients in the reference
    order = int(rng.integers(3, 7))
    leaf = int(rng.integers(20, 120)) if adaptive else int(rng.integers(60, 200))
    params = (leaf, O.COMPRESSION_ACA, 10.0 ** -order, 1024)
    t = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid), base_range=0.3, total_sill=0.2), adaptive, sparse, extents=extents,
                  params=F.FmmParams(*params))
    r = O.FmmTree(pts, order, kid, adaptive, sparse, extents, O.FmmParams(*params), base_range=0.3, total_sill=0.2)
    inject_product_operators(t, r)
    pool_w = {k: [rng.standard_normal((n, k)) for _ in range(2)] for k in (1, 2, 3)}
    x_few = rng.random((int(rng.integers(1, 60)), d)) * 0.98 + 0.01
    x_many = rng.random((int(rng.integers(300, 2500)), d)) * 0.98 + 0.01
    row_sets = [np.sort(rng.choice(n, int(rng.integers(n // 50 + 1, n // 2)), replace=False)).astype(np.int64) for _ in range(2)]
    poly = rng.standard_normal((n, 4))
    log, bad = [], None
    cur_k, locals_stored = 0, False
    for c in range(n_calls):
        op = "set_weights" if c == 0 else ["set_weights", "at_sources", "few", "many", "rows", "grads", "locals", "leaves",
                                           "leaves_grads", "fmv", "fmv_rows", "device"][int(rng.integers(0, 12))]
        k = [1, 2, 3][int(rng.integers(0, 3))] if (op in ("set_weights", "device") or rng.integers(0, 10) == 0) else cur_k
        w = pool_w[k][int(rng.integers(0, 2))]
        x = x_few if rng.integers(0, 2) else x_many
        if rng.integers(0, 8) == 0:
            x = x.copy()
            x[rng.choice(len(x), min(3, len(x)), replace=False), 0] = 7.0
        rows = row_sets[int(rng.integers(0, 2))]
        wl = np.concatenate([pool_w[1][int(rng.integers(0, 2))][:, 0], rng.standard_normal(4)])   # N + basis_size rows (rbf.rs:1344)
        compare = True
        if op == "set_weights":
            ft, fr = (lambda: (t.set_weights(w), np.zeros(0))[1]), (lambda: (r.set_weights(w), np.zeros(0))[1])
        elif op == "at_sources":
            ft, fr = (lambda: t.evaluate(w, pts.copy())), (lambda: r.evaluate(w, pts.copy()))
        elif op in ("few", "many"):
            ft, fr = (lambda: t.evaluate(w, x)), (lambda: r.evaluate(w, x))
        elif op == "rows":
            ft, fr = (lambda: t.evaluate(w, pts[rows])), (lambda: r.evaluate(w, pts[rows]))
        elif op == "grads":
            ft = lambda: np.concatenate([a.ravel() for a in t.evaluate_with_gradients(w, x)])
            fr = lambda: np.concatenate([np.asarray(a).ravel() for a in r.evaluate_with_gradients(w, x)])
        elif op == "locals":
            ft, fr = (lambda: (t.set_local_coefficients(w), np.zeros(0))[1]), (lambda: (r.set_local_coefficients(w), np.zeros(0))[1])
        elif op == "leaves":
            ft, fr = (lambda: t.evaluate_leaves(w, x)), (lambda: r.evaluate_leaves(w, x))
        elif op == "leaves_grads":
            ft = lambda: np.concatenate([a.ravel() for a in t.evaluate_leaves_with_gradients(w, x)])
            fr = lambda: np.concatenate([np.asarray(a).ravel() for a in r.evaluate_leaves_with_gradients(w, x)])
        elif op == "fmv":
            ft = lambda: t.fast_matrix_vector_product(wl, basis_size=4, polynomial_matrix=poly, nugget=0.25)
            fr = lambda: O.fast_matrix_vector_product(r, wl, basis_size=4, polynomial_matrix=poly, nugget=0.25)
        elif op == "fmv_rows":
            ft = lambda: t.fast_matrix_vector_product(wl, basis_size=4, target_indices=rows, polynomial_matrix=poly)
            fr = lambda: O.fast_matrix_vector_product(r, wl, basis_size=4, target_indices=rows, polynomial_matrix=poly)
        else:  # device-resident vectors: the reference's sequence is set_weights + evaluate at the source rows
            dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()

            def ft(dw=dw, k=k):
                out = torch.zeros((k, n), dtype=torch.float64, device="cuda")
                t.matvec_device(dw.data_ptr(), n, k, out.data_ptr(), n, True)
                return out.cpu().numpy().T

            def fr(w=w):
                r.set_weights(w)
                return r.evaluate(w, pts.copy())
        takes_weights_of_k = op not in ("set_weights", "fmv", "fmv_rows", "device")
        if takes_weights_of_k and k != cur_k:
            compare = False                 # another column count than set_weights': open in the reference, refused by the product
        leaves_call = op in ("leaves", "leaves_grads")
        if leaves_call and not locals_stored:
            compare = False                 # outside the header's contract
        if compare:
            yt, et = run(ft)
            yr, er = run(fr)
        else:
            yt, et = run(ft)                # made on the product only (the oracle would index past its arrays or read stale L)
            yr, er = None, None
        entry = {"op": op, "k": k, "compared": compare, "refused": et is not None}
        if compare:
            if (et is None) != (er is None) or (et is not None and et[0] != er[0]):
                bad = dict(entry, why="one side refused the call and the other did not, or they name different rows", product=et, oracle=er)
            elif et is None:
                e = relerr(yt, yr)
                entry["err"] = e
                if not (e < TOL):
                    bad = dict(entry, why="values differ")
        ok_call = et is None
        if op in ("set_weights", "device", "fmv", "fmv_rows") and ok_call:
            cur_k = k if op in ("set_weights", "device") else 1
        if op == "locals":
            locals_stored = compare and ok_call
        elif not leaves_call and not (takes_weights_of_k and k != cur_k and not ok_call):
            locals_stored = False
        log.append(entry)
        if bad:
            break
    res = {"sequence": s, "d": d, "n": n, "kernel": kid, "order": order, "leaf": leaf, "adaptive": adaptive, "sparse": sparse,
           "extents": extents is not None,
           "calls": [e["op"] + ("!" if e["refused"] else "") + ("" if e["compared"] else "~") for e in log],
           "max_err": max([e.get("err", 0.0) for e in log] or [0.0]), "ok": bad is None}
    if bad:
        res["failure"] = bad
        fails += 1
    print(json.dumps(res), flush=True)
    del t, r
print(json.dumps({"sequences": n_seq, "calls_per_sequence": n_calls, "failures": fails}))
sys.exit(1 if fails else 0)
