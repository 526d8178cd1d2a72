#!/usr/bin/env python3
"""r04: k_spmm_ilv16p (one wave per chunk, pipelined across its Y blocks) against k_spmm_ilv16 (one wave per Y block) -- the solve must
not differ in a single bit.  Each setting in a process of its own on the LAB build (tests/_env_worker.py).
usage: python scripts/pipe_bits.py [fixture ...]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["fd_16x16_2d", "stencil:40:40:16:16:5:7:5", "stencil:33:29:16:16:3:11:9"]
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for name in names:
        for hv in (1, 0):
            res = {}
            for tag, env in (("pipe", dict(TFQMRGPU_PIPE=1, TFQMRGPU_PIPE_MIN=1)), ("wave_per_block", dict(TFQMRGPU_PIPE=0))):
                out = os.path.join(tmp, tag + ".npz")
                e = dict(os.environ, TFQMRGPU_HASHV3=str(hv), **{k: str(v) for k, v in env.items()})
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_env_worker.py"), out, name, "z", "1e-9"], env=e, capture_output=True, text=True, timeout=900)
                if r.returncode:
                    print(name, tag, "FAILED", r.stdout[-1500:], r.stderr[-3000:]); bad += 1; break
                res[tag] = np.load(out)
            if len(res) < 2: continue
            a, b = res["pipe"], res["wave_per_block"]
            same = (int(a["status"]) == int(b["status"]) and int(a["iterations"]) == int(b["iterations"]) and np.array_equal(a["history"], b["history"])
                    and float(a["residual"]) == float(b["residual"]) and np.array_equal(a["X"], b["X"]))
            print("%-32s hashv3 %d: status %d | %d, iterations %d | %d, residual %.3e | %.3e, max|dX| %.2e -> %s" % (
                name, hv, int(a["status"]), int(b["status"]), int(a["iterations"]), int(b["iterations"]), float(a["residual"]), float(b["residual"]),
                float(np.abs(a["X"] - b["X"]).max()), "bit-identical" if same else "DIFFERENT"), flush=True)
            bad += 0 if same else 1
sys.exit(1 if bad else 0)
