#!/bin/bash
source scripts/gpu_steps.sh
cat > gpurun_out/onecol.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tfqmrgpu_amd as T
from tfqmrgpu_amd import problems as PR
p1 = PR.stencil_2d(256, 256, 16, 16, 1, seed=11)
s1 = T.Solver()
s1.create_plan(p1)
v1 = s1.plan_view()
s1.set_buffer(nbytes=s1.buffer_size(16, 16, "z"))
s1.set_matrix("A", p1.A)
s1.set_matrix("X", np.random.default_rng(2).uniform(-1, 1, (p1.nnzbX, 16, 16)) + 0j)
nP1, nY1 = v1["nPairs"], p1.nnzbX
b1 = (nP1 + 2 * nY1) * 2 * 16 * 16 * 8 + 4 * (nY1 + 1) + 8 * nP1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s1.apply_operator(2); torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0.record(); s1.apply_operator(20); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
S = p1.nnzbX * 2 * 256 * 8
ms = min(ts) - 2 * S / 5.5e9 / 20
print("ORDER=%s: one-column multiply %.4f ms (min of 5 x 20) = %.0f GB/s = %.3f of 8 TB/s" % (os.environ.get("TFQMRGPU_ORDER", "default"), ms, b1 / ms / 1e6, b1 / ms / 1e6 / 8000), flush=True)
PY
step 300 onecol_d.txt python gpurun_out/onecol.py
step 300 onecol_0.txt env TFQMRGPU_ORDER=0 python gpurun_out/onecol.py
step 300 onecol_g1.txt env TFQMRGPU_ORDER_G=1 python gpurun_out/onecol.py
tail -1 gpurun_out/onecol_d.txt gpurun_out/onecol_0.txt gpurun_out/onecol_g1.txt
