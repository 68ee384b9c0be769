"""The upward pass of a partition (option (ii) with an all-reduce; the passes being split are
bbfmm.rs:666-772), checked on the host without a device: every rank's plan is walked with point COUNTS in place of
multipoles (P2M -> points of the leaf, M2M -> sum over the plan's children).  Summed over the ranks the coarse prefix
must hold the true counts (what the all-reduce delivers), and every cell a rank reads above the coarse level must
hold its true count already -- for every world size, on a mixed-level tree, for every coarse level."""
import os

import numpy as np
import pytest

import ferreus_rbf_rs_amd as F


def _true_counts(tree, d):
    keys, leaf = tree.cells()
    ptr, _ = tree.leaf_sources()
    cnt = np.diff(ptr).astype(np.int64)
    level = (keys & np.uint64(0x7FFF)).astype(np.int64)
    index = {int(k): i for i, k in enumerate(keys)}
    for i in np.argsort(-level, kind="stable"):                     # deepest first: children before parents
        lv = int(level[i])
        if lv == 0:
            continue
        pk = ((int(keys[i]) >> 15) >> d) << 15 | (lv - 1)          # morton.rs: parent = code >> d, level - 1
        cnt[index[pk]] += cnt[i] if not leaf[index[pk]] else 0
    return cnt, level


@pytest.mark.parametrize("world,coarse", [(2, None), (3, None), (8, None), (8, 1), (5, 2), (4, 0)])
def test_partial_upward_plans_sum_to_the_whole_upward_pass(world, coarse):
    rng = np.random.default_rng(17)
    pts = np.vstack([rng.random((30000, 3)), np.clip(rng.normal(size=(15000, 3)) * 0.04 + 0.6, 0.0, 0.999)])
    tree = F.FmmTree(pts, 3, F.KernelParams(F.FmmKernelType.LinearRbf), True, True,
                     params=F.FmmParams(40, 2, 1e-3, 1024), host_only=True)
    st = tree.stats()
    assert st.depth >= 5 and st.n_w > 0
    truth, level = _true_counts(tree, 3)
    assert truth[0] == len(pts)
    if coarse is None:
        os.environ.pop("BBFMM_PART_COARSE_LEVEL", None)
    else:
        os.environ["BBFMM_PART_COARSE_LEVEL"] = str(coarse)
    try:
        total = None
        owned_rows = 0
        for r in range(world):
            tree.set_partition(r, world)
            owned_rows += len(tree.partition_rows())
            counts, reads, info = tree.debug_partition_upward_counts()
            lc, n_coarse = int(info[0]), int(info[1])
            assert lc == (coarse if coarse is not None else lc) and lc <= st.depth - 1
            assert n_coarse == (int((level <= lc).sum()) if lc > 0 else 0)
            assert tree.partition_coarse_count() == n_coarse * 32          # n = 27 -> n_pad = 32
            fine = (level > lc) & (reads == 1)
            assert np.array_equal(counts[fine], truth[fine]), (world, r)    # complete where the rank reads
            part = np.where(level <= lc, counts, 0) if lc > 0 else np.zeros_like(counts)
            assert (part[level <= lc] >= 0).all() or lc == 0
            total = part if total is None else total + part
            if coarse is None and world > 1:
                assert info[2] < st.n_leaves                                # not the whole tree's leaves
        assert owned_rows == len(pts)
        if lc > 0:
            sel = (level <= lc) & (level >= 1)
            assert np.array_equal(total[sel], truth[sel]), world            # the all-reduce completes the coarse levels
    finally:
        os.environ.pop("BBFMM_PART_COARSE_LEVEL", None)
        tree.set_partition(0, 1)
