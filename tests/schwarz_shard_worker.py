"""One rank of the sharded-preconditioner test (tests/test_gpu_schwarz_sharded.py starts `world` of these on the one GPU
of the box, exchange over gloo): the Schwarz preconditioner with its factors sharded over the ranks
(bbfmm_schwarz_create_sharded) against the unsharded one built by the same process -- one apply bit for bit, then the
whole FGMRES solve.  Handles are BBFMM_FLAG_DETERMINISTIC so that the ranks' replicated partial products agree bit for
bit and every rank takes the same branches (the collectives need the ranks in step).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n = int(sys.argv[1])
    kid = int(sys.argv[2])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd import solvers as S
    from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
    rng = np.random.default_rng(91)
    pts = rng.random((n, 3))
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.5 * pts[:, 2] ** 2
    tree = F.FmmTree(pts, 6, F.KernelParams(F.KernelType(kid), base_range=0.3, total_sill=0.3), True, True, deterministic=True)
    st = InterpolantSettings(kid, 3, base_range=0.3, total_sill=0.3)
    prm = DDMParams(1024, 0.5, 0.125, 2000)      # 60k points: 60k -> 7.5k -> 938 (coarse), ~128 and ~16 domains
    whole = SchwarzPreconditioner(tree, pts, st, prm)
    shard = SchwarzPreconditioner(tree, pts, st, prm, shard_group=True)
    m = st.basis_size
    r = rng.standard_normal(n + m)
    r[n:] = 0.0
    z0 = whole(r)
    z1 = shard(r)
    z2 = shard(r)                                            # again: the exchange buffer is reused
    own = [shard.domains_owned(lv) for lv in range(shard.num_levels)]
    op = S.RbfSystemOperator(tree, m, whole.monomial_matrix, 0.0)
    rhs = np.concatenate([vals, np.zeros(m)])
    x0, h0 = S.fgmres(op, rhs, whole, None, 4, 5, S.FittingAccuracy(1e-6))
    x1, h1 = S.fgmres(op, rhs, shard, None, 4, 5, S.FittingAccuracy(1e-6))
    print(json.dumps({"rank": rank, "world": world, "levels": shard.num_levels,
                      "apply_equal": bool(np.array_equal(z0, z1) and np.array_equal(z1, z2)),
                      "apply_max_diff": float(np.abs(z0 - z1).max()), "apply_norm": float(np.abs(z0).max()),
                      "owned": own, "factor_bytes_whole": whole.factor_bytes(), "factor_bytes_shard": shard.factor_bytes(),
                      "iterations": [len(h0), len(h1)], "history_equal": bool([a[1] for a in h0] == [a[1] for a in h1]),
                      "solution_equal": bool(np.array_equal(x0, x1)), "final_residual": float(h1[-1][1])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
