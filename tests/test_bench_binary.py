"""The compiled bench_tfqmrgpu (tfqmrgpu_amd/csrc/bench_tfqmrgpu.cpp): the reference's benchmark driver
(tfQMRgpu/source/bench_tfqmrgpu.cu:442-590) as a C++ caller of the staged C-ABI, same arguments and result lines."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "bench_tfqmrgpu")
GOLD = os.path.join(ROOT, "tests", "golden")


def test_binary_is_built_and_prints_usage():
    assert os.path.exists(EXE), "make -C tfqmrgpu_amd/csrc builds it"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Usage:" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f", "z"])
def test_multi_mode_on_the_reference_plan_file(prec):
    r = subprocess.run([EXE, "multi", os.path.join(GOLD, "plan_unordered.14-287-16.gz"), prec, "3", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    dev = float(re.search(r"# GPU maxdev (\S+)", r.stdout).group(1))
    assert dev <= (1e-4 if prec == "f" else 1e-11)
    # 50 526 block products of 16x16x16 complex: the flop count of bench_tfqmrgpu.cu:335
    tflop = float(re.search(r"# GPU performed (\S+) T", r.stdout).group(1))
    assert tflop == pytest.approx(6 * 50526 * 8.0 * 16 ** 3 * 1e-12, abs=6e-4)
    assert re.search(r"# GPU performance \(lm,ln,tune\)=\( 16, 16,0\) is +\S+ G[fF]lop/sec", r.stdout)
    # this build's extension: the CPU re-computation timed beside it, and the roofline fraction
    assert re.search(r"# CPU performance \(host re-computation in double, \d+ threads\) is +\S+ GFlop/sec", r.stdout)
    m = re.search(r"# MI355X roofline: (\S+) GB/s of compulsory bytes \((\S+) of 8 TB/s HBM\), (\S+) T[fF]lop/s \((\S+) of the (\S+) T[fF]lop/s matrix peak\): (\S+)-bound, fraction (\S+)", r.stdout)
    assert m and float(m.group(5)) == (157.3 if prec == "f" else 78.6) and 0 < float(m.group(7)) < 1


@pytest.mark.gpu
def test_tfqmr_mode_matches_the_golden_solve():
    g = np.load(os.path.join(GOLD, "fd_16x16_small.npz"))
    r = subprocess.run([EXE, "tfQMR", os.path.join(GOLD, "fd_16x16_small.xml"), "z", "1", "2000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"# GPU converged to (\S+) in (\d+) iterations", r.stdout)
    # the default (hash) shadow vector is not the golden run's glibc sequence: same system, nearly the same count
    assert abs(int(m.group(2)) - int(g["solve_z_iterations"])) <= 3
    assert float(m.group(1)) <= float(g["solve_z_threshold"])


@pytest.mark.gpu
def test_tfqmr_mode_in_mixed_precision():
    # `bench_tfqmrgpu tfQMR file m`: the reference's driver feeds float arrays and passes 'm' on (bench_tfqmrgpu.cu:105,140-153,573);
    # its library then refuses the solve (status 16, tfqmrgpu.cu:42-44) -- here it converges to the DOUBLE threshold of the file
    r = subprocess.run([EXE, "tfQMR", os.path.join(GOLD, "fd_16x16_small.xml"), "m", "1", "2000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "requested precision= 'm'" in r.stdout
    m = re.search(r"# GPU converged to (\S+) in (\d+) iterations", r.stdout)
    assert m and float(m.group(1)) <= 1e-9 and 14 <= int(m.group(2)) <= 60          # the sum of the float iterations of the refinement
    py = subprocess.run([os.sys.executable, "-m", "tfqmrgpu_amd.bench_tfqmrgpu", "tfQMR", os.path.join(GOLD, "fd_16x16_small.xml"), "m", "1", "2000"],
                        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert py.returncode == 0, py.stdout + py.stderr
    assert re.search(r"# GPU converged to (\S+) in (\d+) iterations", py.stdout).groups() == m.groups()


@pytest.mark.gpu
def test_tfqmr_mode_extension_lines_and_one_rank_through_rccl():
    # iterations per second and the fused multiply against the HBM roof; `--gpus 1`: a child process per GPU, RCCL communicator
    g = np.load(os.path.join(GOLD, "fd_16x16_2d.npz"))
    plain = subprocess.run([EXE, "tfQMR", os.path.join(GOLD, "fd_16x16_2d.xml"), "z", "1", "2000"], capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stdout + plain.stderr
    assert re.search(r"# GPU iterations per second: \S+ \(9 block columns x 16 right-hand sides", plain.stdout)
    m = re.search(r"# MI355X roofline: fused multiply (spmm_v4_dot|spmm_v5_nrm_dot), (\d+) launches of (\S+) ms: (\S+) GB/s of algorithmic bytes = (\S+) of 8 TB/s HBM", plain.stdout)
    # (the launches of steady iterations: the first iteration of a solve skips the vectors that are zero there and is kept apart)
    assert m and int(m.group(2)) == int(g["solve_z_iterations"]) - 1 and 0 < float(m.group(5)) < 1
    ranks = subprocess.run([EXE, "tfQMR", os.path.join(GOLD, "fd_16x16_2d.xml"), "z", "1", "2000", "--gpus", "1"],
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert ranks.returncode == 0, ranks.stdout + ranks.stderr
    a = re.search(r"# GPU converged to (\S+) in (\d+) iterations", plain.stdout)
    b = re.search(r"# GPU converged to (\S+) in (\d+) iterations", ranks.stdout)
    assert a.groups() == b.groups()                                  # the all-reduced stopping test decides the same
    assert re.search(r"# 1 GPUs: \S+ T[fF]lop in \S+ seconds \(slowest rank\) = \S+ T[fF]lop/s aggregate, %s iterations" % a.group(2), ranks.stdout)
