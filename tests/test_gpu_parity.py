"""-m gpu: the HIP path (through the C ABI) against the oracle on the same seeded inputs.
Tolerances: FP64, different LU (multifrontal without pivoting, product-form panels vs
SuperLU/LAPACK in the oracle) => relative 2-norm difference of one ApplyInverse
<= 1e-10 for exact (levels=0) configurations and <= 1e-8 multilevel."""
import os

import numpy as np
import pytest

from common import problem, xml_params, oracle_prec, product_prec, rel_diff
from oracle import krylov

pytestmark = pytest.mark.gpu

CASES = [
    # eq, n, sx, levels, cx, tol
    ("Laplace", 8, 4, 0, -1, 1e-10),
    ("Laplace", 16, 4, 1, -1, 1e-10),
    ("Laplace", 16, 4, 2, 2, 1e-10),
    ("Laplace", 32, 4, 2, -1, 1e-10),   # reference testSuite/integration_tests/threeD1.xml
    ("Laplace", 32, 8, 1, -1, 1e-10),
    ("Stokes-C", 8, 4, 0, -1, 1e-9),
    ("Stokes-C", 16, 8, 0, -1, 1e-8),
]


@pytest.mark.parametrize("eq,n,sx,levels,cx,tol", CASES)
def test_apply_inverse_matches_oracle(gpu_lib, eq, n, sx, levels, cx, tol):
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx), gpu_lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx)
    assert [s[1] for s in P.level_sizes()][: len(O.level_sizes())] == [s[1] for s in O.level_sizes()]
    rng = np.random.default_rng(42)
    for _ in range(2):
        b = rng.uniform(-1, 1, A.shape[0])
        assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol
    # multivector (column-major, ld = n) and repeatability
    B = rng.uniform(-1, 1, (A.shape[0], 3))
    X = P.ApplyInverse(B)
    for k in range(3):
        assert rel_diff(X[:, k], O.apply_inverse(B[:, k])) < tol
    assert np.array_equal(P.ApplyInverse(B[:, 0]), X[:, 0])  # bitwise reproducible
    assert P.NumApplyInverse() >= 4 and P.NumCompute() == 1


SKEW = [
    (8, 4, 0, -1, {}, 1e-9),
    (16, 8, 1, -1, {}, 1e-8),
    (16, 4, 2, 2, {"Eliminate Velocities Together": False}, 1e-8),   # stokes2_3D.xml shape
    (32, 8, 1, -1, {}, 1e-8),                                        # stokes1_3D.xml shape (32^3)
]


@pytest.mark.parametrize("n,sx,levels,cx,extra,tol", SKEW)
def test_stokes_skew_matches_oracle(gpu_lib, n, sx, levels, cx, extra, tol):
    A, tv = problem("Stokes-C", n)
    P = product_prec(A, tv, xml_params("Stokes-C", n, sx, levels, cx, "Skew Cartesian", extra=extra), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, sx, levels, cx, partitioner="Skew Cartesian",
                    link_velocities=extra.get("Eliminate Velocities Together", True))
    assert [s[1] for s in P.level_sizes()][: len(O.level_sizes())] == [s[1] for s in O.level_sizes()]
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol
    # divergence-free property (reference testSuite/integration_tests/integration_tests.cpp:453-484)
    b0 = b.copy(); b0[3::4] = 0.0
    assert np.abs((A @ P.ApplyInverse(b0))[3::4]).max() <= 1e-8 * np.abs(b0).max() * A.shape[0]


def test_stokes_gmres_iterations_match_oracle(gpu_lib):
    """stokes1_3D.xml shape at 16^3: right-preconditioned GMRES, tol 1e-8; reference target <= 130."""
    n, sx = 16, 8
    A, tv = problem("Stokes-C", n)
    P = product_prec(A, tv, xml_params("Stokes-C", n, sx, 1, partitioner="Skew Cartesian"), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, sx, 1, partitioner="Skew Cartesian")
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, A.shape[0]); b = A @ x
    _, its_o, res_o = krylov.gmres(lambda v: A @ v, b, O.apply_inverse, tol=1e-8, maxit=200)
    _, its_p, res_p = krylov.gmres(lambda v: A @ v, b, P.ApplyInverse, tol=1e-8, maxit=200)
    assert abs(its_p - its_o) <= 1 and its_p <= 130 and res_p < 1e-7


def test_device_resident_vectors(gpu_lib):
    import torch
    A, tv = problem("Laplace", 16)
    P = product_prec(A, tv, xml_params("Laplace", 16, 4, 1), gpu_lib)
    b = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    x_host = P.ApplyInverse(b)
    x_dev = P.ApplyInverse(torch.from_numpy(b).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(x_dev.cpu().numpy(), x_host)
    y = P.MatVec(torch.from_numpy(b).cuda()).cpu().numpy()
    assert rel_diff(y, A @ b) < 1e-14


def test_exact_inverse_levels0(gpu_lib):
    """levels=0 => ApplyInverse(K x) = x (reference testSuite/unit_tests/HYMLS_Preconditioner.cpp:247-276)."""
    A, tv = problem("Laplace", 16)
    P = product_prec(A, tv, xml_params("Laplace", 16, 4, 0), gpu_lib)
    x = np.random.default_rng(2).uniform(-1, 1, A.shape[0])
    assert np.abs(P.ApplyInverse(A @ x) - x).max() < 1e-10


def test_krylov_iteration_count_matches_oracle(gpu_lib):
    """threeD1.xml: 3D Laplace 32^3, sx=4, levels=2, CG tol 1e-10: <= 35 iterations in the
    reference; GPU path and oracle must need the same number."""
    A, tv = problem("Laplace", 32)
    P = product_prec(A, tv, xml_params("Laplace", 32, 4, 2), gpu_lib)
    O = oracle_prec(A, tv, "Laplace", 32, 4, 2)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, A.shape[0]); b = A @ x
    x0 = rng.uniform(-1, 1, A.shape[0])
    _, its_o, res_o = krylov.pcg(lambda v: A @ v, b, O.apply_inverse, tol=1e-10, maxit=100, x0=x0)
    _, its_p, res_p = krylov.pcg(lambda v: A @ v, b, P.ApplyInverse, tol=1e-10, maxit=100, x0=x0)
    assert its_p == its_o and its_p <= 35 and res_p <= 1e-9


def test_set_matrix_recompute(gpu_lib):
    """SetMatrix with the same pattern + Compute reuses the symbolic work (reference
    src/HYMLS_Preconditioner.hpp:244-254, testSuite/unit_tests/HYMLS_Preconditioner.cpp:106-124)."""
    A, tv = problem("Laplace", 16)
    P = product_prec(A, tv, xml_params("Laplace", 16, 4, 1), gpu_lib)
    A2 = A * 2.0
    P.SetMatrix(A2)
    assert not P.IsComputed()
    P.Compute()
    O = oracle_prec(A2, tv, "Laplace", 16, 4, 1)
    b = np.random.default_rng(4).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-10
    assert P.NumInitialize() == 1 and P.NumCompute() == 2


def test_errors(gpu_lib):
    import hymls_amd
    A, tv = problem("Laplace", 8)
    P = hymls_amd.Preconditioner(A, xml_params("Laplace", 8, 4, 0), testVector=tv, lib=gpu_lib)
    with pytest.raises(hymls_amd.HymlsError) as e:
        P.ApplyInverse(np.ones(A.shape[0]))
    assert e.value.code == -1  # "The preconditioner has not yet been computed."
    assert P.Apply(None, None) == -1 and P.SetUseTranspose(True) == -1 and P.Condest() == -1.0
    # 3D Stokes-C with the Cartesian partitioner: singular pressure-tube blocks are reported
    A, tv = problem("Stokes-C", 8)
    P = hymls_amd.Preconditioner(A, xml_params("Stokes-C", 8, 4, 1), testVector=tv, lib=gpu_lib)
    with pytest.raises(hymls_amd.HymlsError) as e:
        P.Compute()
    assert e.value.code == -4


def test_big_front_path_matches_oracle():
    """Force the multi-workgroup ("big front") factor/solve kernels (MFMA FP64 Schur update,
    row-chunked forward, split-k backward) on small problems by lowering the work threshold in a
    fresh process, and compare with the oracle."""
    import os, subprocess, sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = r'''
import numpy as np, sys
sys.path.insert(0, "tests")
from common import problem, xml_params, oracle_prec, product_prec, rel_diff
import hymls_amd
lib = hymls_amd.load_library()
for (eq, n, sx, levels, part, tol) in [("Laplace", 16, 4, 1, "Cartesian", 1e-10), ("Laplace", 32, 8, 1, "Cartesian", 1e-10),
                                      ("Stokes-C", 16, 8, 1, "Skew Cartesian", 1e-8), ("Stokes-C", 16, 4, 2, "Skew Cartesian", 1e-8)]:
    A, tv = problem(eq, n)
    cx = 2 if levels == 2 else -1
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, part), lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx, partitioner=part)
    b = np.random.default_rng(12).uniform(-1, 1, A.shape[0])
    d = rel_diff(P.ApplyInverse(b), O.apply_inverse(b))
    print(eq, n, sx, levels, d)
    assert d < tol, (eq, n, sx, levels, d)
print("BIG-OK")
'''
    env = dict(os.environ, HYMLS_MI_BIG_FLOPS="3000")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "BIG-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world,args", [
    (2, ("Stokes-C", 32, 16, 16, 8, 1, -1, "Skew Cartesian")),
    (4, ("Stokes-C", 32, 32, 16, 4, 2, 2, "Skew Cartesian")),
])
def test_sharded_matches_single_gpu(gpu_lib, world, args):
    """the sharded path on the real kernels: `world` ranks share this box's one MI355X (RCCL cannot run two
    ranks on one device, so the transport stages through gloo here; on a multi-GPU node the same callbacks
    run dist.all_to_all_single on RCCL).  Assembled sharded ApplyInverse == one-rank ApplyInverse."""
    from test_sharded import run_worker
    res = run_worker(world, args, "gpu", 29540 + world, timeout=900, env_extra={"HYMLS_TEST_BORDER": "1"})
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-10 and res["border_err"] < 1e-9        # plain and bordered ApplyInverse
    assert res["repeat_diff"] == 0.0 and res["recompute_diff"] < 1e-12


@pytest.mark.gpu
def test_sharded_x_periodic_matches_single_gpu(gpu_lib):
    """the same on a periodic grid (x-periodic channel): the halo of a rank's box wraps around
    (tests/test_periodic.py::test_sharded_x_periodic_matches_one_rank is the CPU twin)"""
    from test_sharded import run_worker
    res = run_worker(2, ("Stokes-C", 32, 16, 16, 8, 1, -1, "Skew Cartesian"), "gpu", 29548, timeout=900, env_extra={"HYMLS_TEST_PERIODIC": "x"})
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-10 and res["matvec_err"] < 1e-13
    assert res["repeat_diff"] == 0.0 and res["recompute_diff"] < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("small_rows", ["256", "16"])
def test_merged_level_solve_path_gpu(gpu_lib, monkeypatch, small_rows):
    """k_lvl_fwd / k_lvl_bwd (solve of subdomains too large for the fused kernel), forced on a small problem:
    whole-front tasks and 64-row tile tasks against the oracle."""
    monkeypatch.setenv("HYMLS_MI_NO_FUSED_SOLVE", "1")
    monkeypatch.setenv("HYMLS_MI_LVL_SMALL_ROWS", small_rows)
    A, tv = problem("Stokes-C", 16)
    P = product_prec(A, tv, xml_params("Stokes-C", 16, 8, 1, partitioner="Skew Cartesian"), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", 16, 8, 1, partitioner="Skew Cartesian")
    b = np.random.default_rng(12).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-9


@pytest.mark.gpu
def test_nonsymmetric_navier_stokes_like_gpu(gpu_lib):
    """BASELINE configs[3] in the small: Stokes3D + linearised convection (nonsymmetric F-matrix), GPU path vs
    oracle, and right-preconditioned GMRES needs the same number of iterations with both."""
    from common import add_convection
    n = 16
    A, tv = problem("Stokes-C", n)
    A = add_convection(A, n, re=80.0)
    P = product_prec(A, tv, xml_params("Stokes-C", n, 8, 1, partitioner="Skew Cartesian"), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, 8, 1, partitioner="Skew Cartesian")
    rng = np.random.default_rng(21)
    b = rng.uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-8
    x = rng.uniform(-1, 1, A.shape[0]); rhs = A @ x
    _, its_o, _ = krylov.gmres(lambda v: A @ v, rhs, O.apply_inverse, tol=1e-8, maxit=250)
    _, its_p, res_p = krylov.gmres(lambda v: A @ v, rhs, P.ApplyInverse, tol=1e-8, maxit=250)
    assert abs(its_p - its_o) <= 1 and res_p < 1e-7


@pytest.mark.gpu
def test_darcy3d_saddle_point_gpu(gpu_lib):
    from oracle import galeri
    n = 16
    A = galeri.darcy3d(n, n, n, 1.0, -1.0)
    tv = galeri.create_testvector(A)
    P = product_prec(A, tv, xml_params("Stokes-C", n, 8, 1, partitioner="Skew Cartesian"), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, 8, 1, partitioner="Skew Cartesian")
    b = np.random.default_rng(22).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-8


def full_size_properties(gpu_lib, problem, n, levels, re=0.0, max_its=400, sizes=None, sx=8, keep=False, restart=None):
    """Size-independent properties at a BASELINE size the oracle cannot reach: the operator is linear, reproducible
    bit for bit, maps pressure-free right-hand sides to divergence-free velocities (reference
    integration_tests.cpp:453-484, 1e-8), and right-preconditioned GMRES on the device (BaseSolver semantics,
    reference src/HYMLS_BaseSolver.cpp:309-397: zero initial guess, b = K x_ex) reaches a relative residual of 1e-8."""
    import torch
    import hymls_amd
    rp, ci, va = hymls_amd.generate_problem(problem, n, n, n, re=re, lib=gpu_lib)
    tv = hymls_amd.generate_testvector(rp, ci, va, lib=gpu_lib)
    P = hymls_amd.Preconditioner((rp, ci, va), xml_params("Stokes-C", n, sx, levels, partitioner="Skew Cartesian"), testVector=tv, lib=gpu_lib)
    P.Compute()
    if sizes:
        assert [s[1] for s in P.level_sizes()] == sizes
    N = rp.size - 1
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    b1 = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b2 = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    x1, x2 = P.ApplyInverse(b1).clone(), P.ApplyInverse(b2).clone()
    x12 = P.ApplyInverse(0.75 * b1 - 2.5 * b2)
    torch.cuda.synchronize()
    assert torch.isfinite(x12).all()
    assert float((x12 - (0.75 * x1 - 2.5 * x2)).norm() / x12.norm()) < 1e-10          # linear
    assert torch.equal(P.ApplyInverse(b1), x1)                                            # reproducible
    b0 = b1.clone(); b0[3::4] = 0.0
    y = P.MatVec(P.ApplyInverse(b0))
    torch.cuda.synchronize()
    assert float(y[3::4].abs().max()) <= 1e-8 * N                                        # divergence-free velocities
    # (not a contraction: ||b - K P b|| >> ||b|| for right-hand sides with a divergence part also with the oracle)
    x_ex = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    rhs = P.MatVec(x_ex).clone()
    S = hymls_amd.Solver(P, P, {"Krylov Method": "GMRES", "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": max_its, "Num Blocks": restart or max_its,
                                                                               "Maximum Restarts": 40}})   # (restart: the Krylov basis of 67 M-vectors is 0.5 GB per block)
    xs = S.ApplyInverse(rhs)
    its = S.getNumIter()
    del S
    true_rel = float((rhs - P.MatVec(xs)).norm() / rhs.norm())
    print("GMRES(%s %d^3 re=%g, Number of Levels=%d): %d iterations, true relative residual %.2e" % (problem, n, re, levels, its, true_rel))
    assert its < max_its and true_rel < 1e-7
    if keep:
        return its, P, b1, x1
    return its


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,levels,re", [("Stokes", 48, 1, 0.0), ("Cavity", 48, 1, 375.0), ("Darcy", 48, 1, 0.0), ("Stokes", 32, 2, 0.0)])
def test_values_against_the_compiled_cpu_oracle(gpu_lib, kind, n, levels, re):
    """ApplyInverse VALUES (not only iteration counts) of the multilevel method at sizes the numpy oracle cannot reach
    in test time: the GPU path against the compiled CPU oracle (oracle/cpu_oracle.py, pinned to the numpy oracle by
    tests/test_cpu_oracle.py) on the same matrix and right-hand sides.  48^3 (442 k DoF) two-level for the three
    BASELINE problem families (cavity: the cell Peclet number of Re = 1000 on 128^3), 32^3 three-level (cx = 2)."""
    import hymls_amd
    from oracle import galeri, cpu_oracle
    A = {"Stokes": lambda: galeri.stokes3d(n, n, n), "Darcy": lambda: galeri.darcy3d(n, n, n, 1.0, -1.0),
         "Cavity": lambda: galeri.oseen3d(n, n, n, re)}[kind]()
    rp, ci, va = hymls_amd.generate_problem(kind, n, n, n, re=re, lib=gpu_lib)
    assert np.array_equal(rp, A.indptr) and np.array_equal(ci, A.indices) and np.array_equal(va, A.data)
    tv = galeri.create_testvector(A)
    sx, cx = (8, -1) if levels == 1 else (4, 2)
    P = product_prec(A, tv, xml_params("Stokes-C", n, sx, levels, cx=cx, partitioner="Skew Cartesian"), gpu_lib)
    from oracle.partition import Params
    p = Params(nx=n, ny=n, nz=n, sx=sx, cx=cx, levels=levels, equations="Stokes-C", partitioner="Skew Cartesian").finalize()
    O = cpu_oracle.Preconditioner(A, p, testvector=tv, nthreads=max(1, len(os.sched_getaffinity(0)))).compute()
    assert O.flags == 0 and [s[1] for s in O.level_sizes()] == [s[1] for s in P.level_sizes()]
    rng = np.random.default_rng(41)
    for _ in range(2):
        b = rng.uniform(-1, 1, A.shape[0])
        assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-8


@pytest.mark.gpu
def test_full_size_properties_128(gpu_lib):
    """BASELINE configs[1]/[2] family: Stokes3D 128^3, 3-level, Skew, sx = 8 (8.4 M DoF).  configs[1] itself (128^3,
    TWO-level) is covered in value by test_values_against_the_compiled_cpu_oracle at 48^3 and at its size -- incl. its
    216 k-unknown last-level direct solve -- by test_full_size_cavity_re1000_128 (2-level) and below."""
    full_size_properties(gpu_lib, "Stokes", 128, 2, sizes=[8388608, 216096, 468])
    full_size_properties(gpu_lib, "Stokes", 128, 1, sizes=[8388608, 216096])          # configs[1] as written


@pytest.mark.gpu
def test_full_size_properties_256_stokes_and_forced_sharded_rccl(gpu_lib, monkeypatch):
    """BASELINE configs[2] at its size on one MI355X: Stokes3D 256^3 (67 108 864 DoF), 3-level, Skew Cartesian, sx = 8 --
    the workload of bench.py's default line.  (a) the size-independent properties (linear, bit-reproducible, divergence-
    free to 1e-8, device GMRES to 1e-8; reference integration_tests.cpp:453-484,486-677), level sizes asserted;
    (b) the SAME problem through the sharded code path on the library's built-in RCCL transport (one rank that is forced
    to shard exchanges every halo / hand-off segment with itself through ncclSend/ncclRecv groups and the self-copy):
    ApplyInverse equals the unsharded one to 1e-10.  N > 1 ranks need N GPUs (the driver's scaling run)."""
    import torch
    import hymls_amd
    from hymls_amd.dist import RcclComm
    n = 256
    its, P0, b, x0 = full_size_properties(gpu_lib, "Stokes", n, 2, sizes=[67108864, 1716288, 3528], keep=True, restart=100, max_its=600)
    assert its <= 300
    x0 = x0.clone()
    del P0
    torch.cuda.empty_cache()      # (the Krylov basis of the GMRES run: 54 GB in torch's cache)
    monkeypatch.setenv("HYMLS_MI_FORCE_SHARDED", "1")
    comm = RcclComm(0, lib=gpu_lib)
    prm = xml_params("Stokes-C", n, 8, 2, partitioner="Skew Cartesian")
    P = hymls_amd.Preconditioner(None, prm, lib=gpu_lib, comm=comm, rank_grid=(1, 1, 1))
    assert P.CommSelfTest() == 0
    req = P.RequiredRows()
    rows = hymls_amd.generate_problem("Stokes", n, n, n, gids=req, lib=gpu_lib)
    P.SetMatrixRows(req, rows)
    P.SetTestVector(hymls_amd.generate_testvector_rows(req, *rows))
    del rows
    P.Compute()
    assert [s[1] for s in P.level_sizes()] == [67108864, 1716288, 3528]
    owned = torch.from_numpy(P.OwnedRows().astype(np.int64)).cuda()
    assert owned.numel() == 4 * n ** 3 and int(torch.unique(owned).numel()) == owned.numel()
    xs = P.ApplyInverse(b[owned].contiguous())
    torch.cuda.synchronize()
    err = float((xs - x0[owned]).norm() / x0.norm())
    print("256^3 forced-sharded (built-in RCCL transport) vs unsharded: relative difference %.2e" % err)
    assert err < 1e-10
    P.close()
    comm.close()


@pytest.mark.gpu
def test_full_size_properties_256_darcy(gpu_lib):
    """BASELINE configs[4] at its size on one MI355X: GaleriExt Darcy3D 256^3 (a = 1, b = -1), 3-level"""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    full_size_properties(gpu_lib, "Darcy", 256, 2, sizes=[67108864, 1716288, 3528], restart=100, max_its=600)


@pytest.mark.gpu
def test_full_size_cavity_re1000_128(gpu_lib):
    """BASELINE configs[3] at its size: Navier-Stokes-like Jacobian at Re = 1000 on 128^3 (cell Peclet numbers up to 2.5:
    the velocity blocks are not diagonally dominant), one MI355X, 2-level and 3-level."""
    full_size_properties(gpu_lib, "Cavity", 128, 1, re=1000.0, max_its=500, sizes=[8388608, 216096])
    full_size_properties(gpu_lib, "Cavity", 128, 2, re=1000.0, max_its=500, sizes=[8388608, 216096, 468])


@pytest.mark.gpu
def test_full_size_darcy_128(gpu_lib):
    """BASELINE configs[4] family (GaleriExt Darcy3D, a = 1, b = -1) at 128^3 = one eighth of the 256^3 problem
    (test_full_size_properties_256_darcy runs that one)."""
    full_size_properties(gpu_lib, "Darcy", 128, 2, sizes=[8388608, 216096, 468])


@pytest.mark.gpu
def test_sharded_transport_over_rccl_single_rank(gpu_lib):
    """the RCCL transport itself (dist.all_to_all_single on the library's stream through
    torch.cuda.ExternalStream, arenas as torch tensors, gloo side group for the host exchanges): one rank that
    is forced to take the sharded code path and exchanges with itself.  Multi-rank RCCL needs one GPU per
    rank and is exercised by bench.py --gpus N on a multi-GPU node."""
    from test_sharded import run_worker
    res = run_worker(1, ("Stokes-C", 32, 32, 32, 4, 2, 2, "Skew Cartesian"), "gpu-nccl", 29551, timeout=600,
                     env_extra={"HYMLS_MI_FORCE_SHARDED": "1"})
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-10 and res["repeat_diff"] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("Stokes-C", 32, 32, 32, 4, 2, 2, "Skew Cartesian"), ("Laplace", 32, 32, 32, 4, 1, -1, "Cartesian")])
def test_builtin_rccl_transport_single_rank(gpu_lib, case):
    """the library's own transport (hymls_mi_set_comm_rccl: ncclSend / ncclRecv groups on the handle's stream, no
    Python frame inside ApplyInverse): one rank forced onto the sharded code path, every exchange is a self-exchange
    through the transport (ncclCommInitRank with one rank, device-to-device self segment, staged host exchanges).
    Results against the one-rank handle; a sharded K x and SetMatrix + recompute included."""
    from test_sharded import run_worker
    res = run_worker(1, case, "gpu-rccl", 29553, timeout=600, env_extra={"HYMLS_MI_FORCE_SHARDED": "1"})
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-10 and res["repeat_diff"] == 0.0 and res["matvec_err"] < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,levels,cx,part,env", [
    ("Laplace", 16, 4, 2, 2, "Cartesian", {}),
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", {}),
    ("Stokes-C", 32, 4, 2, 2, "Skew Cartesian", {}),
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", {"HYMLS_MI_NO_FUSED_SOLVE": "1"}),                       # task kernels on level 0
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", {"HYMLS_MI_NO_FUSED_SOLVE": "1", "HYMLS_MI_LVL_SMALL_ROWS": "32"}),   # 64-row tiles
])
def test_multivector_apply_inverse_gpu(gpu_lib, monkeypatch, eq, n, sx, levels, cx, part, env):
    """several right-hand sides in one ApplyInverse (reference src/HYMLS_MatrixBlock.cpp:335-344, Preconditioner.cpp:
    997-1002): the multi-vector kernels (factor panels read once per group of up to 4 columns) against the oracle and
    against the single-vector kernels, nvec = 2, 3, 4, 5, 8; host blocks and device tensors."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx=cx, partitioner=part), gpu_lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx=cx, partitioner=part)
    rng = np.random.default_rng(8)
    for nvec in (2, 3, 4, 5, 8):
        B = rng.uniform(-1, 1, (A.shape[0], nvec))
        X = P.ApplyInverse(B)
        for j in range(nvec):
            x1 = P.ApplyInverse(B[:, j].copy())
            assert rel_diff(X[:, j], x1) < 1e-12                      # (summation order differs from the single-vector kernels)
        for j in (0, nvec - 1):
            assert rel_diff(X[:, j], O.apply_inverse(B[:, j])) < 1e-8
        Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()       # (nvec, n) row-major = column-major (n, nvec)
        Xd = P.ApplyInverse(Bd)
        torch.cuda.synchronize()
        assert np.array_equal(Xd.cpu().numpy().T, X)


@pytest.mark.gpu
def test_unstable_pivot_free_factorisation_is_reported(gpu_lib):
    """the subdomain LU runs without numerical pivoting (the reference runs KLU with pivot tolerance 0 on its F-matrix
    ordering, src/HYMLS_SparseDirectSolver.cpp:244-254); element growth above 1e8 is detected on the device and Compute
    returns -4 instead of handing out a silently inaccurate factor.  Matrix: the Laplace couplings with a diagonal of
    1e-11 (every first pivot of a leaf is tiny); a benign shift of the same pattern factors fine."""
    import hymls_amd
    import scipy.sparse as sp
    n = 8
    A, tv = problem("Laplace", n)
    off = A - sp.diags(A.diagonal())
    prm = xml_params("Laplace", n, 4, 1)
    bad = (off + 1e-11 * sp.identity(A.shape[0])).tocsr()
    with pytest.raises(hymls_amd.HymlsError) as e:
        product_prec(bad, tv, prm, gpu_lib)
    # (element growth, or an exactly cancelled pivot further down the same front: both are reported as -4)
    assert e.value.code == -4 and ("unstable" in str(e.value) or "pivot" in str(e.value))
    good = (off - 7.0 * sp.identity(A.shape[0])).tocsr()
    P = product_prec(good, tv, prm, gpu_lib)
    O = oracle_prec(good, tv, "Laplace", n, 4, 1)
    b = np.random.default_rng(2).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,levels,cx,part", [("Stokes-C", 16, 8, 1, -1, "Skew Cartesian"), ("Stokes-C", 32, 4, 2, 2, "Skew Cartesian"),
                                                  ("Laplace", 32, 4, 2, -1, "Cartesian")])
def test_compute_is_bitwise_reproducible(gpu_lib, eq, n, sx, levels, cx, part):
    """Compute twice (and on a second handle): the same bits in ApplyInverse.  The updates of several root fronts to
    one separator block are added in a fixed order (they were concurrent atomic adds in round 1)."""
    A, tv = problem(eq, n)
    prm = xml_params(eq, n, sx, levels, cx=cx, partitioner=part)
    P = product_prec(A, tv, prm, gpu_lib)
    b = np.random.default_rng(13).uniform(-1, 1, A.shape[0])
    x1 = P.ApplyInverse(b)
    for _ in range(2):
        P.Compute()
        assert np.array_equal(P.ApplyInverse(b), x1)
    Q = product_prec(A, tv, prm, gpu_lib)
    assert np.array_equal(Q.ApplyInverse(b), x1)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2])
def test_retain_nodes_gpu(gpu_lib, which):
    from test_hostsim_parity import RETAIN, retain_nodes_case
    retain_nodes_case(gpu_lib, *RETAIN[which])


@pytest.mark.gpu
def test_two_live_handles_alternate(gpu_lib):
    """every handle owns its device context (stream, arenas, profiling marks): two preconditioners alive at the same
    time, applied alternately, give exactly what each gives alone (reference: any number of Preconditioner objects)."""
    A1, tv1 = problem("Laplace", 16)
    A2, tv2 = problem("Stokes-C", 16)
    prm1, prm2 = xml_params("Laplace", 16, 4, 2, cx=2), xml_params("Stokes-C", 16, 8, 1, partitioner="Skew Cartesian")
    b1 = np.random.default_rng(3).uniform(-1, 1, A1.shape[0]); b2 = np.random.default_rng(4).uniform(-1, 1, A2.shape[0])
    P1 = product_prec(A1, tv1, prm1, gpu_lib); x1 = P1.ApplyInverse(b1); del P1
    P2 = product_prec(A2, tv2, prm2, gpu_lib); x2 = P2.ApplyInverse(b2); del P2
    Pa = product_prec(A1, tv1, prm1, gpu_lib)
    Pb = product_prec(A2, tv2, prm2, gpu_lib)
    assert Pa.stream() != Pb.stream()
    Pa.set_profiling(True)
    for _ in range(3):
        assert np.array_equal(Pa.ApplyInverse(b1), x1)
        assert np.array_equal(Pb.ApplyInverse(b2), x2)
    Pb.Compute()                                   # setup of one handle between applies of the other
    assert np.array_equal(Pa.ApplyInverse(b1), x1) and np.array_equal(Pb.ApplyInverse(b2), x2)
    assert Pa.last_apply_seconds(0) > 0 and Pb.last_apply_seconds(0) == 0     # profiling marks are per handle


@pytest.mark.gpu
@pytest.mark.parametrize("re,part,levels,its_max", [("re1000", "Cartesian", 1, 100), ("re1000", "Skew Cartesian", 1, 110), ("re1000", "Cartesian", 2, 120)])
def test_reference_driven_cavity_2d_gpu(gpu_lib, re, part, levels, its_max):
    """the reference's 2D driven-cavity Jacobian at Re 1000 (its own test data) on the GPU path"""
    from test_hostsim_parity import cavity2d_case
    cavity2d_case(gpu_lib, re, part, levels, its_max)


@pytest.mark.gpu
def test_reference_driven_cavity_2d_64_gpu(gpu_lib):
    """the reference's 64x64 Re 1000 system (the reference-pinned robustness check), 3-level, GPU path:
    oracle parity of one ApplyInverse, same GMRES iteration count, the stored solution is reached"""
    from test_hostsim_parity import cavity2d_case
    cavity2d_case(gpu_lib, "re1000", "Cartesian", 2, 250, grid=64)


@pytest.mark.gpu
@pytest.mark.parametrize("levels,tol,re", [(1, 1e-8, 10000.0), (1, 1e-8, 500.0)])
def test_high_reynolds_robustness_gpu(gpu_lib, levels, tol, re):
    """pivot-free interior factorisation where the velocity rows have lost diagonal dominance: the add_convection
    case re = 10000 on 16^3 (cell Reynolds number 625) and the configs[3] generator at a cell Peclet number four
    times that of Re = 1000 on 128^3, GPU path against the oracle (SuperLU with partial pivoting)"""
    from common import add_convection
    from oracle import galeri
    n = 16
    if re > 1000:
        A, tv = problem("Stokes-C", n)
        A = add_convection(A, n, re=re)
    else:
        A = galeri.oseen3d(n, n, n, re); tv = galeri.create_testvector(A)
    P = product_prec(A, tv, xml_params("Stokes-C", n, 8, levels, partitioner="Skew Cartesian"), gpu_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, 8, levels, partitioner="Skew Cartesian")
    b = np.random.default_rng(21).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol


@pytest.mark.gpu
def test_c_abi_from_plain_c(gpu_lib, tmp_path):
    """include/hymls_mi.h from a C program (gcc, no Python / torch in the process): create, set matrix, Compute,
    ApplyInverse with host buffers, error code before Compute, exactness of the one-level method."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "capi_smoke")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "capi", "capi_smoke.c"), "-o", exe,
                           "-L", os.path.join(root, "hymls_amd"), "-lhymls_mi", "-Wl,-rpath," + os.path.join(root, "hymls_amd"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CAPI_SMOKE" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("nb,nblk", [(1, 3), (10, 5), (64, 7), (88, 3), (100, 3), (160, 3), (191, 4), (392, 3), (791, 2), (1024, 1), (1100, 3), (2050, 1)])
def test_separator_block_inversion_gpu(gpu_lib, nb, nblk):
    """hymls_mi_invert_blocks (Ifpack_DenseContainer::Compute in the reference, SchurPreconditioner.cpp:284-291): orders
    >= 160 take the blocked Gauss-Jordan route (32 pivots per panel, matrix-core update), orders that are no multiple of
    32 included; above 1024 (one panel row per thread, in registers) the panel step works in a global scratch copy.  Checked
    against LAPACK (numpy):
    |X A - I| small, and blocks that NEED the row interchanges (zero diagonal) are inverted as well."""
    A, tv = problem("Laplace", 8)
    P = product_prec(A, tv, xml_params("Laplace", 8, 4, 0), gpu_lib)
    rng = np.random.default_rng(nb)
    B = rng.uniform(-1, 1, (nblk, nb, nb))
    B[0] += nb * np.eye(nb)                               # one diagonally dominant block (the separator blocks are)
    if nblk > 1 and nb > 1:
        B[1][np.arange(nb), np.arange(nb)] = 0.0          # zero diagonal: no LU without interchanges
    if nblk > 2:
        B[2] = 65536.0 * (np.diag(np.full(nb, 6.0)) - np.diag(np.ones(nb - 1), 1) - np.diag(np.ones(nb - 1), -1)) if nb > 1 else B[2]
    X = P.InvertBlocks(B)
    for q in range(nblk):
        ref = np.linalg.inv(B[q])
        cond = np.linalg.cond(B[q])
        assert np.abs(X[q] - ref).max() <= 1e-13 * cond * np.abs(ref).max(), (nb, q, cond)
        assert np.abs(X[q] @ B[q] - np.eye(nb)).max() <= 1e-13 * cond


@pytest.mark.gpu
def test_singular_separator_block_is_reported_gpu(gpu_lib):
    from hymls_amd.api import HymlsError
    A, tv = problem("Laplace", 8)
    P = product_prec(A, tv, xml_params("Laplace", 8, 4, 0), gpu_lib)
    for nb in (40, 200):
        B = np.random.default_rng(1).uniform(-1, 1, (2, nb, nb))
        B[1][:, 7] = 0.0
        with pytest.raises(HymlsError) as e:
            P.InvertBlocks(B)
        assert e.value.code == -4


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,levels,cx", [("Laplace", 16, 4, 1, -1), ("Stokes-C", 16, 8, 1, -1), ("Laplace", 16, 4, 2, 2)])
def test_recompute_keeps_the_coarse_plan_gpu(gpu_lib, eq, n, sx, levels, cx):
    """SetMatrix + Compute with the same pattern: the coarse direct solver keeps its ordering and symbolic factorisation and
    only refactors (CoarseSolver::Compute keeps its Amesos solver in the reference, src/HYMLS_CoarseSolver.cpp:131-152);
    the result must be what a fresh preconditioner for the new matrix gives, also after going back to the first matrix."""
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), gpu_lib)
    b = np.random.default_rng(11).uniform(-1, 1, A.shape[0])
    rng = np.random.default_rng(12)
    A2 = A.copy()
    A2.data = A2.data * (1.0 + 0.05 * rng.uniform(-1, 1, A2.data.size))       # same pattern, other values
    if eq == "Stokes-C":
        A2 = A.copy(); A2.data = A2.data * 1.5
    P.SetMatrix(A2)
    P.Compute()
    F = product_prec(A2, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), gpu_lib)
    assert rel_diff(P.ApplyInverse(b), F.ApplyInverse(b)) < 1e-12
    # back to the first matrix: again a refactorisation
    P.SetMatrix(A)
    P.Compute()
    F = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), gpu_lib)
    assert rel_diff(P.ApplyInverse(b), F.ApplyInverse(b)) < 1e-12
    assert P.NumInitialize() == 1 and P.NumCompute() == 3


@pytest.mark.gpu
def test_darcy_separator_length_16_rounding_level_pressure_diagonals_gpu(gpu_lib):
    """Darcy3D 64^3 with separator length 16, two-level: summed on the GPU, 42 of the 199 pressure diagonals of the reduced
    matrix come out as 1.05e-14 (next to entries of 1e3) instead of the exact zeros a CPU summation gives.  The ordering
    of the last-level solver tells pressures from velocities by their zero diagonal (reference MatrixUtils.cpp:1344-1352),
    so those have to count as zeros -- they once made the pivot-free factorisation grow by 1e13 (Compute returned -4).
    The CPU oracle needs 73 GMRES iterations for this problem."""
    its = full_size_properties(gpu_lib, "Darcy", 64, 1, max_its=100, sx=16, sizes=[4 * 64 ** 3, 3528])
    assert its <= 80


@pytest.mark.gpu
def test_two_pass_separator_transform_fallback_gpu(gpu_lib, tmp_path):
    """Separator blocks whose three vectors exceed the LDS of the fused transform + dropping kernel (order 6000+: separator
    length 32) take the two-sided Householder passes + extraction instead.  Forced here on a small problem (the switch is
    read once per process, so the check runs in a child process) and compared with the oracle."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from common import problem, xml_params, oracle_prec, product_prec, rel_diff
        import hymls_amd
        lib = hymls_amd.load_library()
        A, tv = problem("Stokes-C", 16)
        P = product_prec(A, tv, xml_params("Stokes-C", 16, 4, 2, 2, "Skew Cartesian"), lib)
        O = oracle_prec(A, tv, "Stokes-C", 16, 4, 2, 2, partitioner="Skew Cartesian")
        b = np.random.default_rng(5).uniform(-1, 1, A.shape[0])
        d = rel_diff(P.ApplyInverse(b), O.apply_inverse(b))
        print("rel diff", d)
        assert d < 1e-8
    """ % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, HYMLS_MI_SBLOCK_TWO_PASS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,part", [("Laplace", 16, 4, "Cartesian"), ("Stokes-C", 16, 4, "Cartesian"), ("Stokes-C", 16, 8, "Cartesian"),
                                          ("Stokes-C", 16, 8, "Skew Cartesian"), ("Stokes-C", 12, 2, "Skew Cartesian")])
def test_partition_matches_oracle_gpu(gpu_lib, eq, n, sx, part):
    """the groups the PRODUCT library (libhymls_mi.so, not the host simulator build of the same sources) hands out
    through hymls_mi_get_interior / hymls_mi_get_separator_groups == the oracle's restatement of
    CartesianPartitioner::GetGroups / SkewCartesianPartitioner + HierarchicalMap ownership, bit for bit"""
    import hymls_amd
    from oracle.partition import Params, HierarchicalMap
    A, tv = problem(eq, n)
    P = hymls_amd.Preconditioner(A, xml_params(eq, n, sx, 0, partitioner=part), testVector=tv, lib=gpu_lib)
    P.Initialize()
    hm = HierarchicalMap(Params(nx=n, ny=n, nz=n, sx=sx, levels=0, equations=eq, partitioner=part).finalize())
    assert P.level_sizes()[0][3] == hm.nsd
    for sd in range(hm.nsd):
        assert np.array_equal(P.interior(0, sd), hm.interior[sd])
        groups = P.separator_groups(0, sd)
        assert len(groups) == len(hm.groups[sd])
        for gi, (typ, owned, nodes) in enumerate(groups):
            assert typ == hm.groups[sd][gi][0] and np.array_equal(nodes, hm.groups[sd][gi][1])
            assert owned == (gi in hm.owned[sd])


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,levels,cx,part", [("Laplace", 24, 4, 2, 2, "Cartesian"), ("Stokes-C", 32, 8, 1, -1, "Skew Cartesian")])
def test_device_built_setup_tables_equal_the_host_built_ones(gpu_lib, monkeypatch, eq, n, sx, levels, cx, part):
    """Round 3 moved table construction of Initialize to the device: the members' entry source lists (k_member_sources), the
    A12 / A21 blocks (k_offdiag), the pull tables of the reduced matrix (k_build_pull_tables), and classes are found by
    signature.  HYMLS_MI_HOST_SOURCE_LISTS builds the source lists on the host and switches the signature shortcut off,
    HYMLS_MI_VERIFY_CLASSES builds every pattern and compares: same bits in all three."""
    import hymls_amd
    A, tv = problem(eq, n)
    b = np.random.default_rng(21).uniform(-1, 1, A.shape[0])
    prm = xml_params(eq, n, sx, levels, cx, part)
    out = []
    for env in ({}, {"HYMLS_MI_HOST_SOURCE_LISTS": "1"}, {"HYMLS_MI_VERIFY_CLASSES": "1"}):
        for k in ("HYMLS_MI_HOST_SOURCE_LISTS", "HYMLS_MI_VERIFY_CLASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        P = hymls_amd.Preconditioner(A, prm, testVector=tv, lib=gpu_lib)
        assert P.Initialize() == 0 and P.Compute() == 0
        out.append(P.ApplyInverse(b))
        P.close()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
