#!/usr/bin/env python3
"""Round-3 fuzz on the GPU box: random clouds / orders / kernels / right-hand sides, each checked three ways against
the default handle's device matvec -- (1) a handle with a small M2L budget (batches, target-class groups, rhs chunks),
(2) a W-way partition run rank by rank through the split upward pass (the sum of the partial coarse multipoles stands in
for the all-reduce), (3) a BBFMM_FLAG_DETERMINISTIC handle, twice, bit for bit, (4) -- round 6 -- ONE handle over W logical parts on the device
(device-resident vectors and the unchanged caller's host-buffer sequence) -- at 1e-12 of max |y|.

  python tests/checks/round3_fuzz.py [cases] [seed]      -> JSON lines, last line = summary"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ferreus_rbf_rs_amd as F

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def cloud(kind, n, d=3):
    if d < 3:
        return np.unique(np.vstack([rng.random((n // 2, d)), np.clip(rng.normal(size=(n - n // 2, d)) * 0.05 + 0.3, 0.0, 0.999)]), axis=0)
    if kind == "uniform":
        return rng.random((n, 3))
    if kind == "clustered":
        k = 8
        c, s, which = rng.random((k, 3)), 0.01 + 0.08 * rng.random(k), rng.integers(0, k, n)
        return np.unique(np.clip(c[which] + rng.normal(size=(n, 3)) * s[which, None], 0.0, 0.999), axis=0)
    if kind == "shell":
        v = rng.normal(size=(n, 3))
        return 0.5 + 0.45 * v / np.linalg.norm(v, axis=1, keepdims=True)
    return np.vstack([rng.random((n // 2, 3)), np.clip(rng.normal(size=(n - n // 2, 3)) * 0.05 + 0.3, 0.0, 0.999)])


fails = 0
for case in range(cases):
    kind = ["uniform", "clustered", "shell", "mixed"][int(rng.integers(0, 4))]
    n = int(rng.integers(20_000, 300_000))
    order = int(rng.integers(4, 9))
    kid = [0, 1, 2, 3, 7][int(rng.integers(0, 5))]
    K = int(rng.integers(1, 4))
    leaf = int(rng.integers(30, 257))
    frac = [0.5, 0.12, 0.03][int(rng.integers(0, 3))]
    world = [2, 3, 5][int(rng.integers(0, 3))]
    d = [3, 3, 3, 2, 1][int(rng.integers(0, 5))]
    adaptive, sparse = bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 4) > 0)
    if d < 3:
        n = min(n, 120_000)
    pts = cloud(kind, n, d)
    n = pts.shape[0]
    kp = F.KernelParams(F.KernelType(kid), base_range=0.3, total_sill=0.2)
    par = F.FmmParams(leaf, 2, 10.0 ** -order, 1024)
    os.environ.pop("BBFMM_M2L_CBUF_MB", None)
    mk = lambda **kw: F.FmmTree(pts, order, kp, adaptive, sparse, params=par, **kw)
    # small leaves: one wave each in the reference handle and the library's size rule (workgroup jobs in trees this small) in
    # the handles checked against it, or the other way round (the variable is read when a handle builds its job lists)
    os.environ["BBFMM_P2P_SYM_WAVE_MIN"] = "0" if case % 2 == 0 else ""
    ref_t = mk()
    os.environ["BBFMM_P2P_SYM_WAVE_MIN"] = "" if case % 2 == 0 else "0"
    st = ref_t.stats()
    w = torch.from_numpy(rng.standard_normal((K, n))).cuda()
    ref = torch.zeros_like(w)
    ref_t.matvec_device(w.data_ptr(), n, K, ref.data_ptr(), n, True)
    scale = float(ref.abs().max())
    res = {"case": case, "cloud": kind, "d": d, "adaptive": adaptive, "sparse": sparse, "n": n, "order": order, "kernel": kid, "K": K, "leaf": leaf, "depth": st.depth, "n_w": st.n_w,
           "reference_handle_small_leaves": "wave jobs" if case % 2 == 0 else "size rule"}
    # (1) small budget
    if st.m2l_slots_bytes_per_rhs > 0:
        os.environ["BBFMM_M2L_CBUF_MB"] = "%.6f" % (frac * st.m2l_slots_bytes_per_rhs / 1048576.0)
    t = mk()
    os.environ.pop("BBFMM_M2L_CBUF_MB", None)
    out = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n, True)
    res["batches"] = t.stats().m2l_batches
    res["err_budget"] = float((out - ref).abs().max()) / scale
    # (2) partition of the budgeted handle, rank by rank
    total = None
    for r in range(world):
        t.set_partition(r, world)
        c = torch.zeros((K, max(t.partition_coarse_count(), 1)), dtype=torch.float64, device="cuda")
        t.matvec_partition_upward(w.data_ptr(), n, K, c.data_ptr())
        torch.cuda.synchronize()
        total = c if total is None else total + c
    full = torch.full_like(w, float("nan"))
    scratch = torch.zeros_like(total)
    for r in range(world):
        t.set_partition(r, world)
        o = torch.zeros_like(w)
        t.matvec_partition_upward(w.data_ptr(), n, K, scratch.data_ptr())
        t.matvec_partition_finish(total.data_ptr(), o.data_ptr(), n, True)
        rows = torch.from_numpy(t.partition_rows()).cuda()
        full[:, rows] = o[:, rows]
    res["world"] = world
    res["err_partition"] = float((full - ref).abs().max()) / scale if not bool(torch.isnan(full).any()) else float("nan")
    del t
    # (3) deterministic handle
    dt = mk(deterministic=True)
    o1, o2 = torch.zeros_like(w), torch.zeros_like(w)
    dt.matvec_device(w.data_ptr(), n, K, o1.data_ptr(), n, True)
    dt.matvec_device(w.data_ptr(), n, K, o2.data_ptr(), n, True)
    res["err_deterministic"] = float((o1 - ref).abs().max()) / scale
    res["deterministic_bitwise"] = bool(torch.equal(o1, o2))
    # (4) round 6: ONE handle over `world` logical parts on the device (bbfmm_create_on_devices), device-resident and host buffers
    gt = mk(devices=[0] * world)
    og = torch.zeros_like(w)
    gt.matvec_device(w.data_ptr(), n, K, og.data_ptr(), n, True)
    res["err_group_device"] = float((og - ref).abs().max()) / scale
    wh = np.asfortranarray(w.cpu().numpy().T)
    gt.set_weights(wh)
    yh = gt.evaluate(wh, pts)
    res["err_group_host"] = float(np.abs(yh.T - ref.cpu().numpy()).max()) / scale
    res["group_path"] = int(gt.last_evaluate_path())
    del gt
    ok = (res["err_budget"] < 1e-12 and res["err_partition"] < 1e-12 and res["err_deterministic"] < 1e-12 and res["deterministic_bitwise"]
          and res["err_group_device"] < 1e-12 and res["err_group_host"] < 1e-12 and res["group_path"] == 1)
    res["ok"] = bool(ok)
    fails += 0 if ok else 1
    print(json.dumps(res), flush=True)
    del dt, ref_t
print(json.dumps({"cases": cases, "failures": fails}))
