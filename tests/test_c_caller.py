"""C boundary: a C99 translation unit written against the reference's prototypes (own stream typedef, host arrays,
one-call driver) compiles with gcc against include/tfqmrgpu.h, links to libtfQMRgpu.so and reproduces the
known-answer test of the reference's Julia example (example/tfqmrgpu_Julia_example.jl:117-120)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

LIBDIR = os.path.join(ROOT, "tfqmrgpu_amd", "lib")
SRC = os.path.join(ROOT, "tests", "c_caller", "kat.c")


def _build(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = str(tmp_path / "kat_c")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + LIBDIR, "-ltfQMRgpu", "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe])
    return exe


def test_c_caller_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_caller_known_answer(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_caller: OK" in r.stdout
