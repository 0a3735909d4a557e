"""The Epetra / Ifpack adapter include/hymls_mi_epetra.hpp as code: compiled with g++ against the minimal
Trilinos stand-ins of tests/mock_epetra/ and driven like a Trilinos application would (Ifpack_Preconditioner and
Epetra_Operator pointers only): lifecycle and error codes of the reference, exactness on one level, SetMatrix,
SetBorder + bordered ApplyInverse, a preconditioned CG loop.  Without a GPU the driver links the TEST-ONLY host
simulator of the C ABI; the -m gpu variant links the product library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(tmp_path, libdir, libname):
    exe = str(tmp_path / "adapter_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "mock_epetra"),
                           os.path.join(ROOT, "tests", "mock_epetra", "adapter_driver.cpp"), "-o", exe,
                           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ADAPTER_OK" in out.stdout, out.stdout + out.stderr
    return out.stdout


def test_adapter_overrides_every_virtual_of_the_reference_interface():
    """virtual for virtual against reference src/HYMLS_Preconditioner.hpp:93-188 (+ SetBorder / SetMatrix)"""
    text = open(os.path.join(ROOT, "include", "hymls_mi_epetra.hpp")).read()
    for name in ("SetParameters", "Initialize", "IsInitialized", "Compute", "IsComputed", "Condest", "Apply", "ApplyInverse",
                 "Matrix", "NumInitialize", "NumCompute", "NumApplyInverse", "InitializeTime", "ComputeTime",
                 "ApplyInverseTime", "InitializeFlops", "ComputeFlops", "ApplyInverseFlops", "Print", "SetUseTranspose",
                 "HasNormInf", "NormInf", "Label", "UseTranspose", "Comm", "OperatorDomainMap", "OperatorRangeMap",
                 "SetBorder", "HaveBorder", "SetMatrix"):
        assert (" " + name + "(") in text, name


def test_adapter_compiles_and_runs_against_mock_epetra(tmp_path, hostsim_lib):
    out = build_and_run(tmp_path, os.path.join(ROOT, "tests", "hostsim"), "hymls_mi_hostsim")
    assert "two-level preconditioned CG" in out


@pytest.mark.gpu
def test_adapter_on_the_gpu(tmp_path, gpu_lib):
    build_and_run(tmp_path, os.path.join(ROOT, "hymls_amd"), "hymls_mi")
