"""Worker of tests/test_gpu_ranks.py: one rank of the column-sharded solve with the PRODUCT (libtfQMRgpu.so) on
cuda:0.  Several ranks share the one GPU of the test box; the stopping test is max-reduced over the ranks through
the host callback of tfqmrgpuExt_setReduceCallback (gloo underneath), i.e. the same protocol and the same enqueue
pattern as the RCCL path, which needs one GPU per rank.  Rank 0 gathers the solution and writes argv[1]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, name, prec, tol = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert torch.cuda.is_available(), "no GPU: the product has no CPU fallback"
    torch.cuda.set_device(0)
    import tfqmrgpu_amd as T
    from conftest import load_problem
    pr = load_problem(name)
    sub, xb, bb = T.shard_columns(pr, world, rank)
    calls = [0]

    def reduce_max(ctx, values, n):
        t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        for i in range(n):
            values[i] = float(t[i])
        calls[0] += 1

    with T.Solver() as s:
        s.create_plan(sub)
        s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
        s.set_matrix("A", sub.A)
        s.set_matrix("B", sub.B)
        keep = T.REDUCE_CB(reduce_max)
        assert T.lib.tfqmrgpuExt_setReduceCallback(s.handle, keep, None) == 0
        st = s.solve(tol, 300)
        info, X, hist = s.get_info(), s.get_matrix(), s.bound_history()
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, status=st, iterations=info["iterations"], residual=info["residual"],
                                          history=hist, xb=xb, X=X, calls=calls[0], n_cols=sub.n_cols))
    if rank == 0:
        Xg = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=X.dtype)
        for g in gathered:
            Xg[g["xb"]] = g["X"]
        np.savez(out, X=Xg, status=[g["status"] for g in gathered], iterations=[g["iterations"] for g in gathered],
                 residual=[g["residual"] for g in gathered], history=np.array([g["history"] for g in gathered]),
                 calls=[g["calls"] for g in gathered], n_cols=[g["n_cols"] for g in gathered])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
