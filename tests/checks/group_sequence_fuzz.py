#!/usr/bin/env python3
"""Round-6 fuzz on the GPU box: random SEQUENCES of the evaluator's calls on a device-group handle (logical parts on the one
GPU) against the same sequence on a plain handle.  What a group keeps between calls -- staged weights, which parts hold whole
multipoles, pending upward passes, stored local expansions, cached row splits -- must never show: after every call the two
handles return the same values (1e-11 of the call's largest value) or both refuse the call.

Calls drawn: set_weights, evaluate at the sources / at few targets / at many targets (sharded over the parts) / at rows of the
sources (the unchanged caller's matvec_partial; drawn from a small pool so that second sightings occur), the same with other
weights than set_weights' (the reference's mixture of old multipoles and new near field), evaluate_with_gradients,
set_local_coefficients + evaluate_leaves (with gradients), fast_matrix_vector_product on all rows and on row sets,
matvec_device with device-resident vectors.

  python tests/checks/group_sequence_fuzz.py [sequences] [seed] [calls per sequence]   -> JSON lines, last line = summary"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BBFMM_GROUP_SHARD_MIN"] = "1500"      # targets per part from which a call is sharded (default 16384)
import torch
import ferreus_rbf_rs_amd as F

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
n_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 14
rng = np.random.default_rng(seed)
TOL = 1e-11


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


def both(fa, fb):
    """Run the same call on both handles: (results or None, error parity ok, messages)"""
    out, errs = [], []
    for f in (fa, fb):
        try:
            out.append(f())
            errs.append(None)
        except ValueError as e:  # every refusal of the library surfaces as ValueError in the binding
            out.append(None)
            errs.append((getattr(e, "point_index", None), str(e)[:120]))
    same = (errs[0] is None) == (errs[1] is None) and (errs[0] is None or errs[0][0] == errs[1][0])  # (the same offending row)
    return out, same, errs


fails = 0
for s in range(n_seq):
    d = 3 if rng.integers(0, 5) else 2
    n = int(rng.integers(20_000, 70_000)) if d == 3 else int(rng.integers(8_000, 30_000))
    pts = np.vstack([rng.random((n // 2, d)), np.clip(rng.normal(size=(n - n // 2, d)) * 0.06 + 0.4, 0.0, 0.999)])
    pts = np.unique(pts, axis=0)
    n = len(pts)
    kid = [0, 1, 2, 3][int(rng.integers(0, 4))]
    order = int(rng.integers(4, 8))
    parts = int(rng.integers(2, 6))
    kp = F.KernelParams(F.KernelType(kid), base_range=0.3, total_sill=0.2)
    det = bool(rng.integers(0, 4) == 0)
    one = F.FmmTree(pts, order, kp, True, True, deterministic=det)
    grp = F.FmmTree(pts, order, kp, True, True, devices=[0] * parts, deterministic=det)
    pool_w = {k: [np.asfortranarray(rng.standard_normal((n, k))) for _ in range(2)] for k in (1, 2)}
    x_few = rng.random((int(rng.integers(1, 200)), d)) * 0.98 + 0.01
    x_many = rng.random((1500 * parts + int(rng.integers(0, 4000)), d)) * 0.98 + 0.01
    m_rows = int(rng.integers(max(1024, n // 2048 + 1), n // 2))
    row_sets = [np.sort(rng.choice(n, m_rows, replace=False)).astype(np.int64) for _ in range(2)]
    log, bad = [], None
    locals_stored = False  # the contract of the leaves-only calls (header: "after bbfmm_set_local_coefficients")
    for c in range(n_calls):
        op = ["set_weights", "at_sources", "few", "many", "rows", "grads", "locals", "leaves", "leaves_grads", "fmv", "fmv_rows",
              "device"][int(rng.integers(0, 12))]
        k = 1 if rng.integers(0, 3) else 2
        w = pool_w[k][int(rng.integers(0, 2))]
        x = x_few if rng.integers(0, 2) else x_many
        if rng.integers(0, 8) == 0:  # a target outside the tree: both handles name the same (first) offending row
            x = x.copy()
            x[rng.choice(len(x), min(3, len(x)), replace=False), 0] = 7.0
        rows = row_sets[int(rng.integers(0, 2))]
        call = None
        if op == "set_weights":
            call = lambda t: (t.set_weights(w), np.zeros(0))[1]
        elif op == "at_sources":
            call = lambda t: t.evaluate(w, pts.copy())
        elif op in ("few", "many"):
            call = lambda t: t.evaluate(w, x)
        elif op == "rows":
            call = lambda t: t.evaluate(w, pts[rows])
        elif op == "grads":
            call = lambda t: np.concatenate([a.ravel() for a in t.evaluate_with_gradients(w, x)])
        elif op == "locals":
            call = lambda t: (t.set_local_coefficients(w), np.zeros(0))[1]
        elif op == "leaves":
            call = lambda t: t.evaluate_leaves(w, x)
        elif op == "leaves_grads":
            call = lambda t: np.concatenate([a.ravel() for a in t.evaluate_leaves_with_gradients(w, x)])
        elif op == "fmv":
            call = lambda t: t.fast_matrix_vector_product(pool_w[1][0][:, 0], nugget=0.25)
        elif op == "fmv_rows":
            call = lambda t: t.fast_matrix_vector_product(pool_w[1][1][:, 0], target_indices=rows)
        else:
            dw = torch.from_numpy(np.ascontiguousarray(w.T)).cuda()

            def call(t, dw=dw, k=k):
                out = torch.zeros((k, n), dtype=torch.float64, device="cuda")
                t.matvec_device(dw.data_ptr(), n, k, out.data_ptr(), n, True)
                return out.cpu().numpy()
        (ya, yb), parity, errs = both(lambda: call(one), lambda: call(grp))
        entry = {"op": op, "k": k, "refused": errs[0] is not None}
        leaves_call = op in ("leaves", "leaves_grads")
        if op == "locals":
            locals_stored = errs[0] is None and errs[1] is None
        elif not leaves_call:
            locals_stored = False          # every other call rewrites the multipoles or the expansions
        if leaves_call and not locals_stored:
            # outside the contract: a plain handle serves the call when the last pass happened to leave whole-tree expansions
            # behind (a matvec), a group -- whose parts hold their own subtrees' -- refuses it; neither changes any state
            entry["outside_contract"] = True
        elif not parity:
            bad = dict(entry, why="one handle refused the call and the other did not, or they name different rows", plain=errs[0], group=errs[1])
        elif ya is not None:
            e = relerr(yb, ya)
            entry["err"] = e
            if not (e < TOL):
                bad = dict(entry, why="values differ")
        log.append(entry)
        if bad:
            break
    res = {"sequence": s, "d": d, "n": n, "kernel": kid, "order": order, "parts": parts, "deterministic": det,
           "calls": [e["op"] + ("!" if e["refused"] else "") + ("~" if e.get("outside_contract") else "") for e in log], "max_err": max([e.get("err", 0.0) for e in log] or [0.0]),
           "ok": bad is None}
    if bad:
        res["failure"] = bad
        fails += 1
    print(json.dumps(res), flush=True)
    del one, grp
print(json.dumps({"sequences": n_seq, "calls_per_sequence": n_calls, "failures": fails}))
sys.exit(1 if fails else 0)
