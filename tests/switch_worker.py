"""Child process of tests/test_gpu_switches.py: one mixed-level cloud through the matvec entry point and a
two-rhs evaluate at arbitrary targets, with whatever BBFMM_* switches the parent put in the environment (they
are read once per process).  Saves the results to the .npz named on the command line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, flags = sys.argv[1], sys.argv[2:]
    import ferreus_rbf_rs_amd as F
    rng = np.random.default_rng(2024)
    n = 460000
    pts = np.vstack([rng.random((400000, 3)), np.clip(rng.normal(size=(60000, 3)) * 0.03 + 0.4, 0.0, 0.999)])
    w = rng.standard_normal((n, 2))
    # 40 points per leaf: the uniform part (12 points per cell: no empty cells) sits at level 5 (32^3 cells), whose faces hold runs of 256 same-class
    # cells with equal V-list patterns -- enough for stage-1 boundary variants when one tile qualifies
    tree = F.FmmTree(pts, 6, F.KernelParams(F.FmmKernelType.LinearRbf), True, True,
                     params=F.FmmParams(40, 2, 1e-6, 1024), deterministic="deterministic" in flags)
    st = tree.stats()
    y = tree.fast_matrix_vector_product(w[:, 0].copy())
    y_again = tree.fast_matrix_vector_product(w[:, 0].copy())
    x = pts[rng.choice(n, 5000, replace=False)] * (1.0 - 1e-9)   # inside occupied leaves (the tree is sparse: empty cells do not exist)
    tree.set_weights(w)
    z = tree.evaluate(w, x)
    u = tree.evaluate(w, pts)                                     # the unchanged caller (rbf.rs:1357-1364): targets = sources
    at_sources = int(tree.last_evaluate_at_sources())
    nv, nc = tree.debug_m2l_variants()
    np.savez(out_path, y=y, y_again=y_again, z=z, u=u, at_sources=at_sources, n_w=st.n_w, depth=st.depth,
             on_device=int(tree.tree_built_on_device()), n_variants=nv)


if __name__ == "__main__":
    main()
