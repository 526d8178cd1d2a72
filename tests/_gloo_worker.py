"""Worker of tests/test_multi_gloo.py: one rank of the column-sharded solve on the CPU.
Launched by torch.distributed.run with the gloo backend.  Every rank
  * cuts its shard out of the global problem with the product's tfqmrgpuExt_shardColumns,
  * takes the matching slice of the GLOBAL shadow vector (results must not depend on the rank count),
  * runs the CPU oracle's tfQMR with the stopping test max-reduced over ranks (same protocol as the
    HIP solver: {max tau/|b|^2, any RHS alive} per iteration, {max res^2, any RHS open} per probe).
Rank 0 gathers the solution blocks and writes them to the file given as argv[1]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, name = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import tfqmrgpu_amd as T
    from conftest import load_problem
    from oracle import pyoracle as O
    if name.startswith("cfg4:"):     # BASELINE config 4's pattern (bench.py --workload cfg4) at a reduced grid: cfg4:nx:ncols
        from bench import build_problem
        pr = build_problem(name, 0, 1)[0]
    else:
        pr = load_problem(name)
    sub, xb, bb = T.shard_columns(pr, world, rank)
    E = 2 * pr.LM * pr.LN
    v3 = O.shadow_glibc(pr.nnzbX * E).reshape(pr.nnzbX, E)[xb].reshape(-1)
    calls = [0]

    def reduce_max(ctx, values, n):
        t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        for i in range(n):
            values[i] = float(t[i])
        calls[0] += 1

    tol = float(sys.argv[3])
    st, X, info = O.solve(sub, "z", threshold=tol, max_iterations=300, v3=v3, reduce=reduce_max)
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, status=st, iterations=info["iterations"], residual=info["residual"],
                                          history=info["bound_history"], xb=xb, X=X, calls=calls[0],
                                          first_col=sub.first_col, n_cols=sub.n_cols))
    if rank == 0:
        Xg = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=np.complex128)
        for g in gathered:
            Xg[g["xb"]] = g["X"]
        np.savez(out, X=Xg, status=[g["status"] for g in gathered], iterations=[g["iterations"] for g in gathered],
                 residual=[g["residual"] for g in gathered], history=np.array([g["history"] for g in gathered]),
                 calls=[g["calls"] for g in gathered], n_cols=[g["n_cols"] for g in gathered])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
