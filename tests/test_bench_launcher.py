"""bench.py as the driver starts it: `python bench.py --gpus N` with no launcher around it starts the N ranks itself as child
processes (torch.distributed.run, one rank per GPU, RCCL communicator of the stopping test inside libtfQMRgpu.so) and
relays rank 0's JSON line.  One rank through that path on the one-GPU test box, two ranks where two GPUs are visible."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--workload", "fd2d_16x16_z_small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-multiply", "--multiply-reps", "2"]


def test_one_rank_through_the_launcher_equals_the_plain_run():
    plain = _run(["--gpus", "1"] + SMALL)
    launched = _run(["--gpus", "1", "--launcher"] + SMALL)
    assert plain["config"]["reduce_path"].startswith("none") and launched["config"]["reduce_path"].startswith("rccl")
    assert launched["n_gpus"] == 1 and launched["solve_status"] == plain["solve_status"] == 0
    assert launched["iterations_per_solve"] == plain["iterations_per_solve"]           # the all-reduced stopping test decides the same
    assert launched["residual"] == plain["residual"]
    for k in ("roofline", "roofline_multiply", "roofline_iteration", "kernels"):
        assert k in launched
    assert launched["roofline"]["kernel"] in ("spmm_v4_dot", "spmm_v5_nrm_dot") and 0 < launched["roofline"]["frac"] < 1


def test_two_ranks_when_two_gpus_are_visible():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box; the N > 1 run is the driver's scaling bench")
    two = _run(["--gpus", "2"] + SMALL)
    assert two["n_gpus"] == 2 and two["solve_status"] == 0 and "2 ranks" in two["config"]["reduce_path"]


def test_unknown_workload_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "nonsense"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unknown workload" in (r.stdout + r.stderr)


def test_config4_workload_builds_only_the_ranks_shard():
    """BASELINE config 4 through bench.py's own build_problem (`--workload cfg4[:nx:ncols]`, strong scaling: the block columns of ONE
    system split over the ranks with tfqmrgpuExt_shardColumns) at a reduced grid: every rank gets a contiguous range of block columns,
    the ranges tile the whole, and a rank's solve gives the bits the unsharded solve has for its columns."""
    import numpy as np
    import tfqmrgpu_amd as T
    sys.path.insert(0, ROOT)
    from bench import build_problem
    full, prec, _ = build_problem("cfg4:12:16", 0, 1)
    assert prec == "z" and (full.first_col, full.n_cols) == (0, 16)
    st, X, info = T.solve_problem(full, "z", threshold=1e-9, max_iterations=200)
    assert st == 0
    col_of = full.colIndX
    at = 0
    for rank in range(3):
        pr, _, desc = build_problem("cfg4:12:16", rank, 3)
        assert pr.first_col == at and "split over 3 GPUs" in desc
        at += pr.n_cols
        stp, Xp, ip = T.solve_problem(pr, "z", threshold=1e-9, max_iterations=200)
        mine = (col_of >= pr.first_col) & (col_of < pr.first_col + pr.n_cols)
        assert stp == 0 and Xp.shape[0] == mine.sum()
        assert np.abs(Xp - X[mine]).max() <= 1e-7 * np.abs(X).max()      # (its own stopping test: the same solution, not the same iteration)
    assert at == 16
