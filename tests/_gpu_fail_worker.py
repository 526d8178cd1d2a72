"""Worker of tests/test_gpu_ranks.py::test_a_failing_rank_stops_every_rank: two ranks share cuda:0, columns sharded, the
stopping test max-reduced through the host callback (gloo).  Both ranks solve with a user-defined operator that repeats the
built-in multiply; the operator of rank `argv[2]` returns an error in its call number `argv[3]`.  Every rank must come back
from tfqmrgpu_bsrsv_solve (none may wait in a reduction for ever) with a non-zero status.  Writes one line per rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, bad_rank, bad_call = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert torch.cuda.is_available(), "no GPU: the product has no CPU fallback"
    torch.cuda.set_device(0)
    import tfqmrgpu_amd as T
    from conftest import load_problem
    pr = load_problem("fd_16x16_small")
    sub, xb, bb = T.shard_columns(pr, world, rank)

    def reduce_max(ctx, values, n):
        t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        for i in range(n):
            values[i] = float(t[i])

    with T.Solver() as s:
        s.create_plan(sub)
        view = s.plan_view()
        s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
        s.set_matrix("A", sub.A)
        s.set_matrix("B", sub.B)
        At = sub.A.transpose(0, 2, 1)
        An = torch.from_numpy(np.ascontiguousarray(np.stack([At.real, At.imag], axis=1))).cuda()
        dS = torch.from_numpy(view["starts"].view(np.int32)).cuda()
        dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()
        calls = [0]

        def multiply(y, x, cols, nnzbX, nCols, lm, ln, precision, stream):
            calls[0] += 1
            if rank == bad_rank and calls[0] == bad_call:
                raise RuntimeError("injected operator failure")      # the thunk turns it into status 14
            T._check(T.lib.tfqmrgpuExt_multiply(s.handle, precision.encode(), lm, ln, nnzbX, dS.data_ptr(), dP.data_ptr(),
                                                An.data_ptr(), x, y), "tfqmrgpuExt_multiply")
            return 0.0

        keep = T.REDUCE_CB(reduce_max)
        assert T.lib.tfqmrgpuExt_setReduceCallback(s.handle, keep, None) == 0
        s.set_operator(multiply)
        st = T.lib.tfqmrgpu_bsrsv_solve(s.handle, s.plan, 1e-9, 300)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, int(st), calls[0]))
    if rank == 0:
        with open(out, "w") as f:
            for g in gathered:
                f.write("%d %d %d\n" % g)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
