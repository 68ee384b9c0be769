"""BBFMM_FLAG_DIRECT_SMALL_W_LEAVES -- an extension beyond the reference (off by default): W-list leaves with no more points
than the expansion has nodes become near field.  Host side (no GPU): the list surgery -- every moved (B, w) pair shows up
in both U lists, the remaining W list plus the moved pairs is the reference's W list, X stays the transpose of W."""
import numpy as np

import ferreus_rbf_rs_amd as F


def _pairs(ptr, idx):
    return {(c, int(j)) for c in range(len(ptr) - 1) for j in idx[ptr[c]:ptr[c + 1]]}


def test_small_w_leaves_move_to_the_u_lists_both_ways():
    rng = np.random.default_rng(17)
    pts = np.vstack([rng.random((9000, 3)), np.clip(rng.normal(size=(9000, 3)) * 0.05 + 0.5, 0, 0.999)])
    kp = F.KernelParams(F.KernelType(0))
    par = F.FmmParams(60, F.M2LCompressionType(2), 1e-4, 1024)
    ref = F.FmmTree(pts, 4, kp, True, True, params=par, host_only=True)          # 64 nodes
    ext = F.FmmTree(pts, 4, kp, True, True, params=par, host_only=True, direct_small_w_leaves=True)
    u0, w0, x0 = (_pairs(*ref.interaction_list(k)) for k in "uwx")
    u1, w1, x1 = (_pairs(*ext.interaction_list(k)) for k in "uwx")
    assert len(w0) > 0 and x0 == {(b, a) for a, b in w0}                          # the reference's lists: X = W^T
    moved = w0 - w1
    assert moved and w1 <= w0                                                     # something moved, nothing appeared
    _, leaf = ref.cells()
    ptr, _ = ref.leaf_sources()
    for b, w in moved:                                                            # only small leaves moved ...
        assert leaf[w] and ptr[w + 1] - ptr[w] <= 64
    for b, w in w1:                                                               # ... and all of them
        assert not (leaf[w] and ptr[w + 1] - ptr[w] <= 64)
    assert x1 == {(b, a) for a, b in w1}
    assert u1 == u0 | moved | {(w, b) for b, w in moved}
    for k in "uwx":                                                               # rows stay sorted by cell index
        p, i = ext.interaction_list(k)
        assert all(np.all(np.diff(i[p[c]:p[c + 1]]) > 0) for c in range(len(p) - 1))
    assert _pairs(*ext.interaction_list("v")) == _pairs(*ref.interaction_list("v"))
