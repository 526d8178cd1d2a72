"""Worker of tests/test_gpu_hash_mode.py::test_plain_mode_xcd_mapping: the reference's plan file (BASELINE config 1) through the native-API multiply
in a fresh process, product written to argv[1]; argv[2] (optional): precision z | c.  With TFQMRGPU_* switches in the environment the LAB build is loaded (as tests/_env_worker.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    prec = sys.argv[2] if len(sys.argv) > 2 else "z"
    real = np.float64 if prec == "z" else np.float32
    import torch
    assert torch.cuda.is_available(), "no GPU: the product has no CPU fallback"
    switches = [k for k in os.environ if k.startswith("TFQMRGPU_") and k != "TFQMRGPU_LIB"]
    if switches:
        os.environ["TFQMRGPU_LIB"] = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "libtfQMRgpu_lab.so")
    import tfqmrgpu_amd as T
    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "plan_unordered.14-287-16.gz"), "rt") as f:   # "#nnzb_for_Y_A_X= nY nA nX", then iY iA iX beta
        head = f.readline().split()
        rows = np.loadtxt(f, dtype=np.int64)
    nY, nA, nX = int(head[1]), int(head[2]), int(head[3])
    starts = np.concatenate([[0], np.flatnonzero(np.diff(rows[:, 0])) + 1, [len(rows)]]).astype(np.int32)   # one group per Y block, in file order
    assert len(starts) == nY + 1
    pairs = np.ascontiguousarray(rows[:, 1:3].reshape(-1).astype(np.int32))
    rng = np.random.default_rng(5)
    A = rng.uniform(-1, 1, (nA, 2, 16, 16)).astype(real)
    X = rng.uniform(-1, 1, (nX, 2, 16, 16)).astype(real)
    _dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with T.Solver() as s:
        dA, dX = _dev(A), _dev(X)
        dY = torch.zeros((nY, 2, 16, 16), dtype=torch.float64 if prec == "z" else torch.float32, device="cuda")
        dS, dP = _dev(starts), _dev(pairs)
        assert T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), 16, 16, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr()) == 0
        torch.cuda.synchronize()
        np.savez(out, Y=dY.cpu().numpy(), lib=os.path.basename(T.LIB_PATH))


if __name__ == "__main__":
    main()
