"""Worker of tests/test_gpu_ranks.py::test_a_refusing_rank_ends_a_mixed_precision_solve_on_every_rank: two ranks share cuda:0, columns
sharded, a mixed-precision ('m') plan on both, the reductions through the host callback (gloo).  Rank argv[2] refuses the solve in
the way argv[3] says: `nobuffer` (no work buffer registered) or `operator` (a user-defined operator, which 'm' plans do not take).
Every rank must come back from tfqmrgpu_bsrsv_solve with a non-zero status and after the SAME number of reductions -- a refusing rank
that entered another collective than its peers (the vote of the inner solve, 2 doubles, against the refinement's reduction, 3
doubles) would hang or corrupt them.  Writes one line per rank: rank status reductions sizes..."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, bad_rank, how = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert torch.cuda.is_available(), "no GPU: the product has no CPU fallback"
    torch.cuda.set_device(0)
    import tfqmrgpu_amd as T
    from conftest import load_problem
    pr = load_problem("fd_16x16_small")
    sub, xb, bb = T.shard_columns(pr, world, rank)
    sizes = []

    def reduce_max(ctx, values, n):
        sizes.append(int(n))
        t = torch.tensor([values[i] for i in range(n)] + [float(n)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # (a mismatch of the sizes between the ranks would fail or hang right here)
        for i in range(n):
            values[i] = float(t[i])

    with T.Solver() as s:
        s.create_plan(sub)
        nbytes = s.buffer_size(pr.LM, pr.LN, "m")
        keep = T.REDUCE_CB(reduce_max)
        assert T.lib.tfqmrgpuExt_setReduceCallback(s.handle, keep, None) == 0
        if not (rank == bad_rank and how == "nobuffer"):
            s.set_buffer(nbytes=nbytes)
            s.set_matrix("A", sub.A)
            s.set_matrix("B", sub.B)
        if rank == bad_rank and how == "operator":
            s.set_operator(lambda *a: 0.0)
        st = T.lib.tfqmrgpu_bsrsv_solve(s.handle, s.plan, 1e-9, 300)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, int(st), sizes))
    if rank == 0:
        with open(out, "w") as f:
            for r, st_, sz in gathered:
                f.write("%d %d %d %s\n" % (r, st_, len(sz), ",".join(str(v) for v in sz)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
