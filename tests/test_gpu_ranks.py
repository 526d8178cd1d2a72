"""Multi-rank path of the PRODUCT on the GPU box: 2 and 3 ranks share cuda:0, block columns sharded with
tfqmrgpuExt_shardColumns, stopping test max-reduced over ranks through the host callback (gloo).  The sharded run
must take the iterations of the single-rank run and return bit-identical solution blocks:
columns are independent systems, the default shadow vector is a hash of (column, row, element), every reduction is
in a fixed order.  At most 4 processes touch the GPU.  Run with `pytest -m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest

import tfqmrgpu_amd as T
from conftest import ROOT, load_problem, torchrun

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,name,prec,tol", [(2, "fd_16x16_small", "z", 1e-9), (3, "stencil_8x8", "z", 1e-9),
                                                 (2, "fd_16x16_2d", "c", 1e-4),
                                                 (2, "fd_16x16_2d", "m", 1e-9)])     # mixed precision: the refinement's residual rides the same max-reduction
def test_ranks_sharing_one_gpu(tmp_path, world, name, prec, tol):
    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = torchrun(world) + [
           os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), out, name, prec, repr(tol)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(out)
    pr = load_problem(name)
    with T.Solver() as s:                       # single rank, same settings
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        st = s.solve(tol, 300)
        info, X, hist = s.get_info(), s.get_matrix(), s.bound_history()
    assert st == 0 and list(g["status"]) == [0] * world
    assert list(g["iterations"]) == [info["iterations"]] * world      # same decisions on every rank
    assert sum(g["n_cols"]) == len(np.unique(pr.colIndX)) and min(g["n_cols"]) >= 1
    for h in g["history"]:                                            # the reduced bound is the global one
        assert np.array_equal(h, hist)
    assert max(g["residual"]) == info["residual"]
    assert np.array_equal(g["X"], X)                                  # bit-identical solution blocks
    assert min(g["calls"]) >= info["iterations"]


def test_a_failing_rank_stops_every_rank(tmp_path):
    """a rank whose operator fails completes its slot with the failure marked in the reduced record: every rank stops at that
    slot and returns an error, none waits in an all-reduce for ever (the reduction carries {bound, alive, a rank failed})"""
    out = str(tmp_path / "status.txt")
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = torchrun(2) + [os.path.join(ROOT, "tests", "_gpu_fail_worker.py"), out, "1", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)    # a hang would end here
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = [tuple(int(v) for v in line.split()) for line in open(out)]
    assert len(rows) == 2
    (r0, st0, calls0), (r1, st1, calls1) = sorted(rows)
    assert T.decode(st1)[0] == 14                       # the failing rank returns its operator's status
    assert st0 != 0 and T.decode(st0)[0] == 2           # the other one: stopped by a peer (LAUNCH_FAILED class)
    assert calls1 in (5, 6) and calls0 in (5, 6)        # both stopped in the slot of the failure (2 operator calls per iteration), far from the 300 iterations asked for


@pytest.mark.parametrize("how", ["nobuffer", "operator"])
def test_a_refusing_rank_ends_a_mixed_precision_solve_on_every_rank(tmp_path, how):
    """'m' plans: a rank that cannot start -- no work buffer, or a user-defined operator, which mixed-precision plans refuse --
    takes part in the FIRST collective its peers enter, the refinement's max-reduction of 3 doubles, with "a rank failed" set; every rank
    leaves there: same number and sizes of reductions on both ranks, the refusing rank returns its own status (7 | 19), the other one 2."""
    out = str(tmp_path / "status.txt")
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = torchrun(2) + [os.path.join(ROOT, "tests", "_gpu_mixed_fail_worker.py"), out, "1", how]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)    # a hang would end here
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = sorted(line.split() for line in open(out))
    assert len(rows) == 2
    (r0, st0, n0, sz0), (r1, st1, n1, sz1) = rows
    assert T.decode(int(st1))[0] == (7 if how == "nobuffer" else 19)      # the refusing rank: its own status
    assert int(st0) != 0 and T.decode(int(st0))[0] == 2                   # its peer: stopped by a rank's failure
    assert n0 == n1 == "1" and sz0 == sz1 == "3"                          # ONE collective, the same on both: the refinement's {res^2, not finite, a rank failed}
