"""Worker of tests/test_gpu_hash_mode.py: solves one fixture in a fresh process and writes status, iterations, residual, bound history
and solution to argv[1].  With TFQMRGPU_* tuning switches in the environment the LAB build of the library is loaded
(libtfQMRgpu_lab.so: the same sources compiled with -DTFQ_LAB, tfq_switch.hpp -- the product has its switches frozen and reads
nothing from the environment); without, the product.
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
    switches = [k for k in os.environ if k.startswith("TFQMRGPU_") and k != "TFQMRGPU_LIB"]
    if switches and "TFQMRGPU_LIB" not in os.environ:
        os.environ["TFQMRGPU_LIB"] = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "libtfQMRgpu_lab.so")
    import tfqmrgpu_amd as T
    assert os.path.basename(T.LIB_PATH) == ("libtfQMRgpu_lab.so" if switches else "libtfQMRgpu.so"), T.LIB_PATH
    st, X, info = T.solve_problem(problem(name), prec, threshold=tol, max_iterations=300)
    np.savez(out, status=st, iterations=info["iterations"], residual=info["residual"], history=info["bound_history"], X=X)


if __name__ == "__main__":
    main()
