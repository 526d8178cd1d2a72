"""Worker of tests/test_gpu_switches.py: solves one fixture with the PRODUCT in a fresh process, so that the library
reads the tuning switches of THIS process' environment (they are read once per process), and writes status,
iterations, residual, bound history and solution to argv[1].
argv: out.npz fixture precision threshold [shape for `stencil:` fixtures]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def problem(name):
    from tfqmrgpu_amd import problems as PR
    if name.startswith("stencil:"):    # stencil:nx:ny:LM:LN:ncols:seed:points
        _, nx, ny, lm, ln, nc, seed, points = name.split(":")
        return PR.stencil_2d(int(nx), int(ny), int(lm), int(ln), int(nc), seed=int(seed), points=int(points))
    from conftest import load_problem
    return load_problem(name)


def main():
    out, name, prec, tol = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    import torch
    assert torch.cuda.is_available(), "no GPU: the product has no CPU fallback"
    import tfqmrgpu_amd as T
    st, X, info = T.solve_problem(problem(name), prec, threshold=tol, max_iterations=300)
    np.savez(out, status=st, iterations=info["iterations"], residual=info["residual"], history=info["bound_history"], X=X)


if __name__ == "__main__":
    main()
