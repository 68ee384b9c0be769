"""One rank of the two-process partition test (tests/test_gpu_two_ranks.py starts two of these as child
processes): the REAL partitioned matvec -- own share of the upward pass, all-reduce of the coarse multipoles,
downward + leaf pass of the owned targets, all-gather of the owned potentials (distributed.PartitionedMatvec) --
compared with the unpartitioned product computed by the same process.  Backend from argv[3]: "gloo" (default;
both ranks on the one GPU of the box, the collectives staged through pinned host memory) or "nccl" (RCCL, one
GPU per rank).  Prints one JSON line."""
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
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    import ferreus_rbf_rs_amd as F
    from ferreus_rbf_rs_amd.distributed import PartitionedMatvec
    pts = np.random.default_rng(77).random((n, 3))
    tree = F.FmmTree(pts, 6, F.KernelParams(F.FmmKernelType.CubicRbf), True, True)
    st = tree.stats()
    w = torch.from_numpy(np.random.default_rng(78).standard_normal((k, n))).to(dev)
    ref = torch.zeros_like(w)
    tree.matvec_device(w.data_ptr(), n, k, ref.data_ptr(), n, True)
    tree.set_partition(rank, world)
    rows = tree.partition_rows()
    pm = PartitionedMatvec(tree, n, k, dev)                      # coarse all-reduce + owned-rows all-gather
    cover = pm.check_partition()
    out = torch.full((k, n), float("nan"), dtype=torch.float64, device=dev)
    errs = []
    for _ in range(2):                                            # twice: buffers are reused across steps
        out.fill_(float("nan"))
        torch.cuda.synchronize()
        pm.step(w, out)
        pm.synchronize()
        torch.cuda.synchronize()
        errs.append(float((out - ref).abs().max() / ref.abs().max()))
    print(json.dumps({"rank": rank, "world": world, "owned": int(len(rows)), "cover": bool(cover), "err": max(errs),
                      "n_w": int(st.n_w), "nan_left": bool(torch.isnan(out).any()),
                      "coarse_count": int(pm.count)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
