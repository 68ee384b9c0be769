"""The intermediate of the two M2L stages is bounded (BBFMM_M2L_CBUF_MB): levels -- or groups of a level's target
classes -- go through one buffer batch after batch, a few right-hand sides per pass.  Whatever the cut, the device
results equal the oracle's (1e-11) and the unbounded default path's (1e-12); the reference itself holds no such
intermediate (bbfmm.rs:864-986 multiplies pair by pair), so this is an implementation bound, not an approximation."""
import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, inject_product_operators, relerr
from oracle import bbfmm_oracle as O

pytestmark = pytest.mark.gpu


def _cloud(seed, n):
    rng = np.random.default_rng(seed)
    return np.vstack([rng.random((n, 3)), clustered_points(rng, n // 3, 3)])


@pytest.mark.parametrize("fraction,nrhs", [(0.6, 1), (0.1, 3), (0.02, 5)])
def test_bounded_intermediate_matches_the_oracle(monkeypatch, fraction, nrhs):
    pts = _cloud(41, 90000)
    whole = F.FmmTree(pts, 6, F.KernelParams(F.KernelType(0)), True, True, host_only=True).stats().m2l_slots_bytes_per_rhs
    budget_mb = fraction * whole / 1048576.0                 # a budget of that fraction of all slots of one rhs
    monkeypatch.setenv("BBFMM_M2L_CBUF_MB", "%.6f" % budget_mb)
    t = F.FmmTree(pts, 6, F.KernelParams(F.KernelType(0)), True, True)
    monkeypatch.delenv("BBFMM_M2L_CBUF_MB")
    r = O.FmmTree(pts, 6, 0, True, True, None, None)
    inject_product_operators(t, r)
    w = np.random.default_rng(1).standard_normal((pts.shape[0], nrhs))
    t.set_weights(w)
    r.set_weights(w)
    y, yr = t.evaluate(w, pts), r.evaluate(w, pts)
    st = t.stats()
    assert st.m2l_batches > 1 and st.n_w > 0
    if fraction <= 0.1:
        assert st.m2l_batches > st.depth - 1                       # a level cut into groups of target classes
    assert st.m2l_intermediate_bytes <= max(budget_mb * 1048576 * 1.01, st.m2l_slots_bytes_per_rhs / 8 * 1.3)
    assert st.m2l_intermediate_bytes < st.m2l_slots_bytes_per_rhs * nrhs
    assert relerr(t.debug_get_coefficients("L", nrhs), r.L) < 1e-11
    assert relerr(y, yr) < 1e-11
    # the matvec entry point (unordered near field, fused lists) and a partial matvec (restricted plan, sparse stage 1)
    ym = t.fast_matrix_vector_product(w[:, 0].copy())
    assert relerr(ym, yr[:, 0]) < 1e-11
    idx = np.sort(np.random.default_rng(2).choice(pts.shape[0], 700, replace=False))
    yp = t.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx)
    assert relerr(yp[idx], yr[idx, 0]) < 1e-11 and np.count_nonzero(np.delete(yp, idx)) == 0


def test_bounded_intermediate_partition_equals_the_default_path(monkeypatch):
    """A 3-way partition (split upward pass, restricted plans) under a budget that cuts the finest levels into groups."""
    import torch
    pts = _cloud(43, 120000)
    n = pts.shape[0]
    ref_tree = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(2)), True, True)
    w = torch.from_numpy(np.random.default_rng(3).standard_normal((2, n))).cuda()
    ref = torch.zeros_like(w)
    ref_tree.matvec_device(w.data_ptr(), n, 2, ref.data_ptr(), n, True)
    assert ref_tree.stats().m2l_batches == 1
    monkeypatch.setenv("BBFMM_M2L_CBUF_MB", "%.6f" % (0.05 * ref_tree.stats().m2l_slots_bytes_per_rhs / 1048576.0))
    t = F.FmmTree(pts, 5, F.KernelParams(F.KernelType(2)), True, True)
    monkeypatch.delenv("BBFMM_M2L_CBUF_MB")
    assert t.stats().m2l_batches > t.stats().depth - 1
    out = torch.zeros_like(w)
    t.matvec_device(w.data_ptr(), n, 2, out.data_ptr(), n, True)
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-12
    world = 3
    total = None
    for rank in range(world):
        t.set_partition(rank, world)
        c = torch.zeros((2, t.partition_coarse_count()), dtype=torch.float64, device="cuda")
        t.matvec_partition_upward(w.data_ptr(), n, 2, c.data_ptr())
        torch.cuda.synchronize()
        total = c if total is None else total + c
    full = torch.full_like(w, float("nan"))
    scratch = torch.zeros_like(total)
    for rank in range(world):
        t.set_partition(rank, world)
        o = torch.zeros_like(w)
        t.matvec_partition_upward(w.data_ptr(), n, 2, scratch.data_ptr())
        t.matvec_partition_finish(total.data_ptr(), o.data_ptr(), n, True)
        rows = torch.from_numpy(t.partition_rows()).cuda()
        full[:, rows] = o[:, rows]
    assert not bool(torch.isnan(full).any())
    assert float((full - ref).abs().max() / ref.abs().max()) < 1e-12
