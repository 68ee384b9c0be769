"""First-contact GPU script: MFMA self-test, per-phase parity vs the oracle, timings.
Writes gpurun_out/gpu_first.log.  (Development aid; the real tests live in tests/.)"""
import os, sys, time, traceback, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ferreus_rbf_rs_amd as F
from oracle import bbfmm_oracle as O

os.makedirs("gpurun_out", exist_ok=True)
LOG = open("gpurun_out/gpu_first.log", "w")
def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

def section(fn):
    try:
        fn()
    except Exception:
        log("EXCEPTION in", fn.__name__); log(traceback.format_exc())

def relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

def inject(T, R):
    fac = {}
    ranks = T.m2l_ranks()
    for lvl in range(2, R.depth + 1):
        fac[lvl] = [T.m2l_factors(lvl, ref) for ref in range(ranks.shape[1])]
    R.set_m2l_operators(fac)

def selftest():
    tf, errs = F.mfma_f64_selftest()
    log("MFMA f64 16x16x4 selftest: layout_errors", errs, "peak TFLOP/s", round(tf, 2), F.mfma_f64_selftest.info)

def parity(N, d, kid, order=7, K=1, adaptive=True, sparse=True, extents=None, grads=False, general_targets=0,
           br=1.0, sill=1.0, params=None, seed=1, clustered=False):
    rng = np.random.default_rng(seed)
    if clustered:
        pts = np.clip(rng.normal(size=(N, d)) * 0.07 + 0.5, 0.0, 0.999)
    else:
        pts = rng.random((N, d))
    w = rng.random((N, K))
    kp = F.KernelParams(F.KernelType(kid), base_range=br, total_sill=sill)
    fp = None if params is None else F.FmmParams(*params)
    t0 = time.time()
    T = F.FmmTree(pts, order, kp, adaptive, sparse, extents=extents, params=fp)
    t_build = time.time() - t0
    op = None if params is None else O.FmmParams(*params)
    R = O.FmmTree(pts, order, kid, adaptive, sparse, extents, op, base_range=br, total_sill=sill)
    inject(T, R)
    s = T.stats()
    tag = f"N={N} d={d} kid={kid} p={order} K={K} adaptive={adaptive} sparse={sparse} depth={s.depth} C={s.n_cells} nW={s.n_w}"
    T.set_weights(w); R.set_weights(w)
    eM = relerr(T.debug_get_coefficients('M', K), R.M)
    tp = pts if not general_targets else (rng.random((general_targets, d)) if extents is None else
                                          rng.random((general_targets, d)) * (np.array(extents[d:]) - np.array(extents[:d])) + np.array(extents[:d]))
    if grads:
        y, g = T.evaluate_with_gradients(w, tp); yr, gr = R.evaluate_with_gradients(w, tp)
        eg = relerr(g, gr)
    else:
        y = T.evaluate(w, tp); yr = R.evaluate(w, tp); eg = 0.0
    eL = relerr(T.debug_get_coefficients('L', K), R.L)
    ey = relerr(y, yr)
    idx = rng.choice(len(tp), min(len(tp), 500), replace=False)
    yd = O.dense_sum(kid, br, sill, tp[idx], pts, w)
    ed = relerr(y[idx], yd)
    log(f"{tag}: build {t_build:.2f}s  M {eM:.1e}  L {eL:.1e}  y {ey:.1e}  grad {eg:.1e}  vs dense {ed:.1e}")
    return T, R, w, pts

def parity_suite():
    parity(20000, 3, 0)
    parity(130000, 3, 0, K=2)
    parity(130000, 3, 2, grads=True)
    parity(60000, 3, 1, order=9, clustered=True)
    parity(40000, 3, 3, br=0.5, sill=0.4, general_targets=5000)
    parity(30000, 3, 7)
    parity(30000, 3, 100, br=1.0)
    parity(30000, 3, 101, br=0.3, sill=0.3)
    parity(30000, 2, 0)
    parity(5000, 1, 0)
    parity(20000, 3, 0, sparse=False, extents=[-1, -1, -1, 2, 2, 2], general_targets=3000, K=2)
    parity(30000, 3, 0, adaptive=False)
    parity(20000, 3, 2, params=(256, 1, 1e-7, 1024))
    parity(8000, 3, 2, params=(64, 0, 1e-7, 1024))
    parity(40000, 3, 0, order=5)
    parity(20000, 3, 2, order=11, params=(400, 2, 1e-9, 1024))

def leaves_api():
    rng = np.random.default_rng(5)
    pts = rng.random((10000, 3)) * 2 - 1; w = rng.random((10000, 1))
    T = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, False, extents=[-2, -2, -2, 2, 2, 2])
    R = O.FmmTree(pts, 7, 0, True, False, [-2, -2, -2, 2, 2, 2]); inject(T, R)
    T.set_weights(w); T.set_local_coefficients(w); R.set_weights(w); R.set_local_coefficients(w)
    x = rng.random((1000, 3)) * 4 - 2
    y = T.evaluate_leaves(w, x); yr = R.evaluate_leaves(w, x)
    yg, g = T.evaluate_leaves_with_gradients(w, x); yr2, gr = R.evaluate_leaves_with_gradients(w, x)
    log("evaluate_leaves", relerr(y, yr), "with grads", relerr(yg, yr2), relerr(g, gr))
    try:
        T.evaluate(w, np.array([[0.0, 0.0, 0.0], [10.0, 0.0, 0.0]]))
        log("ERROR: no PointOutsideTree")
    except F.PointOutsideTree as e:
        log("PointOutsideTree ok:", e.point_index)
    # reference unit test bbfmm.rs:1464-1500
    T1 = F.FmmTree(np.array([[0.5]]), 3, F.KernelParams(F.FmmKernelType.LinearRbf), True, False, extents=[0.0, 1.0])
    T1.set_weights(np.array([[1.0]]))
    try:
        T1.evaluate(np.array([[1.0]]), np.array([[0.5], [10.0]]))
        log("ERROR: reference unit test did not raise")
    except F.PointOutsideTree as e:
        log("reference unit test PointOutsideTree index", e.point_index)
    # fast_matrix_vector_product incl. subset, nugget, polynomial
    N = 20000
    pts = rng.random((N, 3)); wfull = rng.random(N + 4)
    P = np.hstack([np.ones((N, 1)), pts])
    T = F.FmmTree(pts, 7, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
    R = O.FmmTree(pts, 7, 0, True, True); inject(T, R)
    r1 = T.fast_matrix_vector_product(wfull, 4, None, P, 0.01)
    r1r = O.fast_matrix_vector_product(R, wfull, 4, None, P, 0.01)
    sub = rng.choice(N, 3000, replace=False)
    r2 = T.fast_matrix_vector_product(wfull, 4, sub, P, 0.01)
    r2r = O.fast_matrix_vector_product(R, wfull, 4, sub, P, 0.01)
    log("fast_matrix_vector_product full", relerr(r1, r1r), "subset", relerr(r2, r2r))

def timing(N, kid=0, order=7, K=1, reps=5):
    import torch
    rng = np.random.default_rng(42)
    pts = rng.random((N, 3))
    t0 = time.time()
    T = F.FmmTree(pts, order, F.KernelParams(F.KernelType(kid)), True, True)
    tb = time.time() - t0
    s = T.stats()
    w = torch.rand((K, N), dtype=torch.float64, device="cuda")
    out = torch.zeros((K, N), dtype=torch.float64, device="cuda")
    T.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, True)
    t0 = time.time()
    for _ in range(reps):
        T.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, False)
    T.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, True)
    dt = (time.time() - t0) / (reps + 1)
    T.set_profiling(True); T.phase_ms(reset=True)
    for _ in range(3):
        T.matvec_device(w.data_ptr(), N, K, out.data_ptr(), N, False)
    ph = {k: v / 3 for k, v in T.phase_ms().items()}
    T.set_profiling(False)
    log(f"TIMING N={N} kid={kid} p={order} K={K}: build {tb:.1f}s depth={s.depth} C={s.n_cells} p2p_pairs={s.p2p_pairs:.3e} "
        f"m2l_flops={s.m2l_flops_k1:.3e} matvec {dt*1e3:.2f} ms ({1/dt:.2f}/s)")
    log("   phases ms:", json.dumps({k: round(v, 3) for k, v in ph.items()}))
    # accuracy spot check vs dense
    idx = rng.choice(N, 300, replace=False)
    yd = O.dense_sum(kid, 1.0, 1.0, pts[idx], pts, w[0].cpu().numpy()[:, None])
    log("   vs dense (300 rows):", relerr(out[0].cpu().numpy()[idx][:, None], yd))
    del T

def timings():
    timing(1_000_000)
    timing(1_000_000, K=4)
    timing(10_000_000)

if __name__ == "__main__":
    which = sys.argv[1:] or ["selftest", "parity", "leaves", "timings"]
    if "selftest" in which: section(selftest)
    if "parity" in which: section(parity_suite)
    if "leaves" in which: section(leaves_api)
    if "timings" in which: section(timings)
    log("DONE")
