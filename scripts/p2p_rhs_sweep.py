#!/usr/bin/env python3
"""Near-field phases of the matvec against the number of right-hand sides (one kernel evaluation per unordered pair
feeds all rhs of a pass: device.hip p2p_sym2_kernel / p2p_sym_kernel / wx_sym_kernel, instances for 1, 2, 4, 8).

  python scripts/p2p_rhs_sweep.py [points=10000000] [kernel=LinearRbf] [rhs list=1,2,3,4,5,8]   -> one JSON line per K

Per K: P2P and P2L (= fused M2P + P2L) ms per matvec from the handle's phase timers, the FP64 lane-instruction rate
(ISA count of bench.pair_issue) against the FMA rate measured in the same process, and the dense-row check."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import ferreus_rbf_rs_amd as F
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    kernel = sys.argv[2] if len(sys.argv) > 2 else "LinearRbf"
    ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4,5,8").split(",")]
    dev = torch.device("cuda", 0)
    pts = np.random.default_rng(42).random((n, 3))
    tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType[kernel], base_range=0.1 if "Spher" in kernel else 1.0,
                                            total_sill=0.1 if "Spher" in kernel else 1.0), True, True)
    st = tree.stats()
    vtf, _ = F.fp64_valu_selftest()
    rate = vtf * 1e12 / 2.0
    for K in ks:
        w = torch.from_numpy(np.random.default_rng(43).random((K, n))).to(dev)
        out = torch.zeros((K, n), dtype=torch.float64, device=dev)
        for _ in range(2):
            tree.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n, sync=True)
        tree.set_profiling(True)
        tree.phase_ms(reset=True)
        steps = 5
        for _ in range(steps):
            tree.matvec_device(w.data_ptr(), n, K, out.data_ptr(), n, sync=True)
        ph, _ = tree.phase_ms(counts=True)
        tree.set_profiling(False)
        rec = {"points": n, "kernel": kernel, "nrhs": K, "P2P_ms": ph["P2P"] / steps, "P2L_ms": ph["P2L"] / steps,
               "M2P_ms": ph["M2P"] / steps}
        for phase in ("P2P", "P2L"):
            pi = bench.pair_issue(st, n, K, kernel, phase)
            if pi and ph[phase] > 0:
                rec[phase + "_frac_of_measured_fma_rate"] = pi[0] * pi[1] / (ph[phase] / steps * 1e-3) / rate
                rec[phase + "_instr_per_evaluation"] = pi[1]
        rec["dense_rows_rel_err"] = bench.dense_rows_err(torch, dev, kernel, 0.1 if "Spher" in kernel else 1.0,
                                                         0.1 if "Spher" in kernel else 1.0, pts, w, out)
        print(json.dumps(rec), flush=True)
        del w, out


if __name__ == "__main__":
    main()
