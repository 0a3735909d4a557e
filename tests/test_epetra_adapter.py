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


# ---- distributed: the adapter under an Epetra_MpiComm, one MPI rank per shard (MPICH of the image)
MPI_PREFIX = "/opt/conda"
MPIEXEC = os.path.join(MPI_PREFIX, "bin", "mpiexec")
needs_mpi = pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(os.path.join(MPI_PREFIX, "lib", "libmpi.so.12"))),
                               reason="no MPICH under /opt/conda")


def mpi_link_args(tmp_path):
    """the image's MPICH sits in a conda prefix whose libstdc++ is older than the system's: link libmpi by path and
    give the executable a private directory with just libmpi and its two Fortran runtime libraries"""
    d = tmp_path / "mpilib"
    d.mkdir(exist_ok=True)
    for f in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        if not (d / f).exists():
            os.symlink(os.path.join(MPI_PREFIX, "lib", f), str(d / f))
    return ["-isystem", os.path.join(MPI_PREFIX, "include"), "-DMPICH_SKIP_MPICXX"], [str(d / "libmpi.so.12"), "-Wl,-rpath," + str(d)]


def build_mpi(tmp_path, libdir, libname, what):
    inc, link = mpi_link_args(tmp_path)
    exe = str(tmp_path / what)
    if what == "adapter_driver_mpi":
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-DHYMLS_MI_HAVE_MPI", "-I", os.path.join(ROOT, "tests", "mock_epetra"),
               os.path.join(ROOT, "tests", "mock_epetra", "adapter_driver_mpi.cpp")]
    else:
        cmd = ["gcc", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "capi", "capi_sharded.c")]
    subprocess.check_call(cmd + ["-I", os.path.join(ROOT, "include")] + inc + ["-o", exe, "-L", libdir, "-l" + libname,
                                                                             "-Wl,-rpath," + libdir] + link + ["-lm"])
    return exe


def run_mpi(exe, ranks, transport, marker, env_extra=None, extra_args=()):
    env = dict(os.environ, OMP_NUM_THREADS="1", HYMLS_MI_HOST_THREADS="2")
    env.update(env_extra or {})
    out = subprocess.run([MPIEXEC, "-n", str(ranks), exe, transport] + list(extra_args), capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0 and marker in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@needs_mpi
@pytest.mark.parametrize("ranks", [2, 4])
def test_distributed_adapter_under_mpiexec(tmp_path, hostsim_lib, ranks):
    """HYMLS_MI::Preconditioner on an Epetra_MpiComm with a row-distributed matrix on Epetra's linear map (mock Epetra_MpiComm /
    Epetra_Import over real MPI): the adapter imports the overlapping rows (hymls_mi_required_rows), shards over the MPI
    transport of include/hymls_mi_mpi.h, maps B / X through hymls_mi_owned_rows.  Sharded ApplyInverse through
    Ifpack_Preconditioner* == the one-rank result to 1e-12; a CG loop that only sees Epetra_Operator converges in the same
    number of iterations; Stokes (Skew) + border.  Reference: src/HYMLS_Preconditioner.hpp:56-84,182-186, src/main.cpp:50-67."""
    exe = build_mpi(tmp_path, os.path.join(ROOT, "tests", "hostsim"), "hymls_mi_hostsim", "adapter_driver_mpi")
    out = run_mpi(exe, ranks, "MPI", "ADAPTER_MPI_OK", extra_args=("periodic",))
    assert "%d ranks (MPI)" % ranks in out and "preconditioned CG through Epetra_Operator" in out
    assert "x-periodic Stokes channel sharded vs one rank" in out       # (periodic grid through the distributed adapter)


@needs_mpi
@pytest.mark.parametrize("ranks,env", [(2, {}), (4, {"HYMLS_MI_HOST_CHUNK_BYTES": "512"})])
def test_plain_c_sharded_caller_under_mpiexec(tmp_path, hostsim_lib, ranks, env):
    """tests/capi/capi_sharded.c: the sharded entry points of hymls_mi.h from plain C over the same MPI transport (the
    second case cuts the setup exchanges into 512-byte point-to-point rounds)"""
    exe = build_mpi(tmp_path, os.path.join(ROOT, "tests", "hostsim"), "hymls_mi_hostsim", "capi_sharded")
    run_mpi(exe, ranks, "MPI", "CAPI_SHARDED_OK", env)


@needs_mpi
@pytest.mark.gpu
def test_distributed_adapter_on_the_gpu(tmp_path, gpu_lib):
    """the same drivers against libhymls_mi.so: two MPI ranks share this box's one MI355X (MPI transport: device segments
    staged through the host, which is what lets two ranks use one card), then the built-in RCCL transport bootstrapped
    over MPI with one rank forced onto the sharded path (RCCL needs one GPU per rank: N > 1 is the 8-GPU node's job)."""
    libdir = os.path.join(ROOT, "hymls_amd")
    exe = build_mpi(tmp_path, libdir, "hymls_mi", "adapter_driver_mpi")
    run_mpi(exe, 2, "MPI", "ADAPTER_MPI_OK")
    out = run_mpi(exe, 1, "RCCL", "ADAPTER_MPI_OK", {"HYMLS_MI_FORCE_SHARDED": "1"})
    assert "1 ranks (RCCL)" in out
    cexe = build_mpi(tmp_path, libdir, "hymls_mi", "capi_sharded")
    run_mpi(cexe, 2, "MPI", "CAPI_SHARDED_OK")
    run_mpi(cexe, 1, "RCCL", "CAPI_SHARDED_OK", {"HYMLS_MI_FORCE_SHARDED": "1"})
