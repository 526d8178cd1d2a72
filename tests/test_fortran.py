"""Fortran boundary: the F90 module `tfqmrgpu` (tfqmrgpu_amd/fortran/tfqmrgpu.F90, same generic names as the
reference module tfqmrgpu_Fortran_module.F90:12-59) on top of the F77-style wrappers in libtfQMRgpu.so."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

FLANG = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
FDIR = os.path.join(ROOT, "tfqmrgpu_amd", "fortran")
LIBDIR = os.path.join(ROOT, "tfqmrgpu_amd", "lib")


def _build(tmp_path):
    if not os.path.exists(FLANG):
        pytest.skip("no Fortran compiler (amdflang) in this image")
    exe = str(tmp_path / "check_module")
    subprocess.check_call([FLANG, "-cpp", "-c", os.path.join(FDIR, "tfqmrgpu.F90"), "-o", str(tmp_path / "tfqmrgpu.o")], cwd=tmp_path)
    subprocess.check_call([FLANG, "-cpp", os.path.join(FDIR, "check_module.F90"), str(tmp_path / "tfqmrgpu.o"),
                           "-L" + LIBDIR, "-ltfQMRgpu", "-Wl,-rpath," + LIBDIR, "-o", exe], cwd=tmp_path)
    return exe


def test_module_and_caller_compile_and_link(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe) and os.path.exists(tmp_path / "tfqmrgpu.mod")


@pytest.mark.gpu
def test_fortran_caller_solves_dense_system(tmp_path):
    # A*X == B checked with matmul inside the program (example/tfqmrgpu_Fortran_example.F90:108-126), 'n' everywhere
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "check_module: OK" in r.stdout
