"""Multi-rank path on the CPU (gloo, world_size 2 and 3): block columns of X/B sharded with the
product's tfqmrgpuExt_shardColumns, stopping test max-reduced over ranks.  The sharded run must take
exactly the iterations of the single-rank run and return bit-identical solution blocks, because the
columns are independent systems and all ranks take the same continue/probe/stop decisions
(SURVEY.md section 8e)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_problem, torchrun


# (8 ranks on the pattern of BASELINE config 4 -- 16x16 complex<double> stencil, block columns split over the ranks, here 6 x 6 rows and
#  16 block columns = 2 per rank -- is the rank count of the driver's scaling run)
@pytest.mark.parametrize("world,name,tol", [(2, "fd_16x16_small", 1e-9), (3, "stencil_8x8", 1e-9), (2, "fd_4x4_2d", 1e-9), (8, "cfg4:6:16", 1e-9)])
def test_sharded_solve_equals_single_rank(oracle, tmp_path, world, name, tol):
    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2" if world < 4 else "1", MASTER_ADDR="127.0.0.1")
    cmd = torchrun(world) + [
           os.path.join(ROOT, "tests", "_gloo_worker.py"), out, name, repr(tol)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(out)
    if name.startswith("cfg4:"):
        sys.path.insert(0, ROOT)
        from bench import build_problem
        pr = build_problem(name, 0, 1)[0]
    else:
        pr = load_problem(name)
    st, X, info = oracle.solve(pr, "z", threshold=tol, max_iterations=300)
    assert st == 0
    assert list(g["status"]) == [0] * world
    assert list(g["iterations"]) == [info["iterations"]] * world      # same decisions on every rank
    assert sum(g["n_cols"]) == len(np.unique(pr.colIndX)) and min(g["n_cols"]) >= 1
    for h in g["history"]:                                            # the reduced bound is the global one
        assert np.array_equal(h, info["bound_history"])
    assert max(g["residual"]) == info["residual"]
    assert np.array_equal(g["X"], X)                                  # bit-identical solution blocks
    assert min(g["calls"]) >= info["iterations"]
