"""One rank of the two-process partition test (tests/test_gpu_two_ranks.py starts two of these as child
processes): the REAL partitioned matvec_device + OwnedRowsExchange over a gloo group, both ranks on the one
GPU of the box (the exchange is staged through pinned host memory), compared with the unpartitioned product
computed by the same process.  Prints one JSON line."""
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
    n, k = int(sys.argv[1]), int(sys.argv[2])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd.distributed import OwnedRowsExchange
    pts = np.random.default_rng(77).random((n, 3))
    tree = F.FmmTree(pts, 6, F.KernelParams(F.FmmKernelType.CubicRbf), True, True)
    st = tree.stats()
    w = torch.from_numpy(np.random.default_rng(78).standard_normal((k, n))).to(dev)
    ref = torch.zeros_like(w)
    tree.matvec_device(w.data_ptr(), n, k, ref.data_ptr(), n, True)
    tree.set_partition(rank, world)
    rows = tree.partition_rows()
    xchg = OwnedRowsExchange(rows, n, k, dev)
    cover = xchg.check_partition()
    stream = torch.cuda.ExternalStream(tree.stream(), device=dev)
    out = torch.full((k, n), float("nan"), dtype=torch.float64, device=dev)
    errs = []
    for _ in range(2):                                            # twice: buffers are reused across steps
        out.fill_(float("nan"))
        torch.cuda.synchronize()
        tree.matvec_device(w.data_ptr(), n, k, out.data_ptr(), n, sync=False)
        with torch.cuda.stream(stream):
            xchg.exchange(out)
        stream.synchronize()
        torch.cuda.synchronize()
        errs.append(float((out - ref).abs().max() / ref.abs().max()))
    print(json.dumps({"rank": rank, "world": world, "owned": int(len(rows)), "cover": bool(cover), "err": max(errs),
                      "n_w": int(st.n_w), "nan_left": bool(torch.isnan(out).any())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
