import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    # the C-ABI library and the oracle are built in-tree by __graft_entry__.build(); make sure they exist
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "libtfQMRgpu.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    exe = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "bench_tfqmrgpu")
    if not (os.path.exists(lib) and os.path.exists(orc) and os.path.exists(exe)):
        g.build()


_build_once()

from tfqmrgpu_amd import problems as PR  # noqa: E402

FD_NAMES = ["fd_8x8_3d", "fd_16x16_2d", "fd_16x16_small", "fd_4x4_2d"]
SYNTHETIC = {
    "julia_kat": lambda: PR.julia_kat(),
    "dense_random": lambda: PR.dense_random(),
    "dense_random_rect": lambda: PR.dense_random(mb=4, LM=4, LN=8, ncols=3, seed=21),
    "stencil_8x8": lambda: PR.stencil_2d(6, 5, 8, 8, 3, seed=3, radius=2.5),
    "stencil_8x32": lambda: PR.stencil_2d(5, 4, 8, 32, 2, seed=4),
}
ALL_NAMES = FD_NAMES + list(SYNTHETIC)


def torchrun(world):
    """one node, `world` ranks; --standalone lets the launcher pick a free rendezvous port itself (no probe-and-hope)"""
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            "--nproc-per-node", str(world)]


def load_problem(name):
    if name in SYNTHETIC:
        return SYNTHETIC[name]()
    return PR.read_xml(os.path.join(GOLDEN, name + ".xml"))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def offset1(pr):
    import tfqmrgpu_amd as T
    return T.Problem(pr.rowPtrA + 1, pr.colIndA + 1, pr.A, pr.rowPtrX + 1, pr.colIndX + 1,
                     pr.rowPtrB + 1, pr.colIndB + 1, pr.B, None, pr.tolerance, 1)


def golden_solves(g):
    """[(precision, threshold, maxit)] recorded in a golden file"""
    out = []
    for prec in "zc":
        if "solve_%s_status" % prec in g:
            out.append((prec, float(g["solve_%s_threshold" % prec]), int(g["solve_%s_maxit" % prec])))
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
