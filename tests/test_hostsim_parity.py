"""-m "not gpu": host logic (partition, ordering, assembly tree, index plans, recursion) driven
end-to-end through the C ABI with the TEST-ONLY device simulator (tests/hostsim), compared with
the oracle.  This validates the integer plans the GPU kernels execute; the kernels themselves
are covered by the -m gpu tests."""
import os

import numpy as np
import pytest

from common import problem, xml_params, oracle_prec, product_prec, rel_diff
from oracle.partition import Params, HierarchicalMap

CASES = [
    ("Laplace", 8, 4, 0, -1, 1e-12),
    ("Laplace", 8, 4, 1, -1, 1e-12),
    ("Laplace", 16, 4, 2, 2, 1e-12),
    ("Laplace", 16, 8, 1, -1, 1e-12),
    ("Stokes-C", 8, 4, 0, -1, 1e-10),
]


@pytest.mark.parametrize("eq,n,sx,levels,cx,tol", CASES)
def test_apply_inverse_matches_oracle(hostsim_lib, eq, n, sx, levels, cx, tol):
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx), hostsim_lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx)
    assert [s[1] for s in P.level_sizes()][: len(O.level_sizes())] == [s[1] for s in O.level_sizes()]
    b = np.random.default_rng(7).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol


@pytest.mark.parametrize("eq,n,sx", [("Laplace", 16, 4), ("Stokes-C", 16, 4), ("Stokes-C", 16, 8)])
def test_partition_matches_oracle(hostsim_lib, eq, n, sx):
    """group lists of the C++ partitioner == oracle restatement (which is pinned by the
    reference's unit-test formulas in test_oracle_pins.py)."""
    A, tv = problem(eq, n)
    import hymls_amd
    P = hymls_amd.Preconditioner(A, xml_params(eq, n, sx, 0), testVector=tv, lib=hostsim_lib)
    P.Initialize()
    hm = HierarchicalMap(Params(nx=n, ny=n, nz=n, sx=sx, levels=0, equations=eq).finalize())
    assert P.level_sizes()[0][3] == hm.nsd
    for sd in range(hm.nsd):
        assert np.array_equal(P.interior(0, sd), hm.interior[sd])
        groups = P.separator_groups(0, sd)
        assert len(groups) == len(hm.groups[sd])
        for gi, (typ, owned, nodes) in enumerate(groups):
            assert typ == hm.groups[sd][gi][0]
            assert np.array_equal(nodes, hm.groups[sd][gi][1])
            assert owned == (gi in hm.owned[sd])


SKEW = [
    # n, sx, levels, cx, extra, tol  (reference testSuite/integration_tests/stokes{0,1,2}_3D.xml shapes)
    (8, 4, 0, -1, {}, 1e-10),
    (8, 4, 1, -1, {}, 1e-10),
    (16, 8, 1, -1, {}, 1e-9),
    (16, 4, 2, 2, {"Eliminate Velocities Together": False}, 1e-9),
]


@pytest.mark.parametrize("n,sx,levels,cx,extra,tol", SKEW)
def test_stokes_skew_matches_oracle(hostsim_lib, n, sx, levels, cx, extra, tol):
    A, tv = problem("Stokes-C", n)
    P = product_prec(A, tv, xml_params("Stokes-C", n, sx, levels, cx, "Skew Cartesian", extra=extra), hostsim_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, sx, levels, cx, partitioner="Skew Cartesian",
                    link_velocities=extra.get("Eliminate Velocities Together", True))
    assert [s[1] for s in P.level_sizes()][: len(O.level_sizes())] == [s[1] for s in O.level_sizes()]
    b = np.random.default_rng(11).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol


@pytest.mark.parametrize("n,sx,nz", [(8, 4, 8), (16, 8, 16), (12, 2, 12), (16, 4, 8)])
def test_skew_partition_matches_oracle(hostsim_lib, n, sx, nz):
    import hymls_amd
    A, tv = problem("Stokes-C", n, nz)
    P = hymls_amd.Preconditioner(A, xml_params("Stokes-C", n, sx, 0, partitioner="Skew Cartesian", nz=nz),
                                 testVector=tv, lib=hostsim_lib)
    P.Initialize()
    hm = HierarchicalMap(Params(nx=n, ny=n, nz=nz, sx=sx, levels=0, equations="Stokes-C",
                                partitioner="Skew Cartesian").finalize())
    assert P.level_sizes()[0][3] == hm.nsd
    for sd in range(hm.nsd):
        assert np.array_equal(P.interior(0, sd), hm.interior[sd])
        groups = P.separator_groups(0, sd)
        assert len(groups) == len(hm.groups[sd])
        for gi, (typ, owned, nodes) in enumerate(groups):
            assert typ == hm.groups[sd][gi][0] and np.array_equal(nodes, hm.groups[sd][gi][1])
            assert owned == (gi in hm.owned[sd])


def test_nonuniform_grid_and_ragged_subdomains(hostsim_lib):
    """nx not a multiple of sx (ragged last subdomains), anisotropic grid."""
    import hymls_amd
    from oracle import galeri
    from oracle.hymls import Preconditioner as OraclePrec
    nx, ny, nz, sx = 10, 8, 6, 4
    A = galeri.laplace3d(nx, ny, nz)
    tv = galeri.create_testvector(A)
    prm = {"Problem": {"Equations": "Laplace", "Dimension": 3, "nx": nx, "ny": ny, "nz": nz},
           "Preconditioner": {"Separator Length": sx, "Number of Levels": 1}}
    P = product_prec(A, tv, prm, hostsim_lib)
    O = OraclePrec(A, Params(nx=nx, ny=ny, nz=nz, sx=sx, levels=1, equations="Laplace").finalize(), testvector=tv).compute()
    b = np.random.default_rng(8).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-12


def test_lifecycle_and_errors(hostsim_lib):
    import hymls_amd
    A, tv = problem("Laplace", 8)
    P = hymls_amd.Preconditioner(A, xml_params("Laplace", 8, 4, 0), testVector=tv, lib=hostsim_lib)
    with pytest.raises(hymls_amd.HymlsError) as e:
        P.ApplyInverse(np.ones(A.shape[0]))
    assert e.value.code == -1
    assert P.Compute() == 0  # auto-initialises (reference Preconditioner.cpp:403-409)
    assert P.IsInitialized() and P.IsComputed() and P.NumInitialize() == 1
    x = np.random.default_rng(9).uniform(-1, 1, A.shape[0])
    assert np.abs(P.ApplyInverse(A @ x) - x).max() < 1e-10  # levels=0 is exact
    A3, tv3 = problem("Stokes-C", 8)
    P = hymls_amd.Preconditioner(A3, xml_params("Stokes-C", 8, 4, 1), testVector=tv3, lib=hostsim_lib)
    with pytest.raises(hymls_amd.HymlsError) as e:
        P.Compute()
    assert e.value.code == -4


@pytest.mark.parametrize("small_rows", ["256", "16"])
def test_merged_level_solve_path(hostsim_lib, monkeypatch, small_rows):
    """the solve path of subdomains too large for the fused kernel (one launch per tree level for all
    classes; whole-front tasks and 64-row tile tasks), forced here on a small problem."""
    monkeypatch.setenv("HYMLS_MI_NO_FUSED_SOLVE", "1")
    monkeypatch.setenv("HYMLS_MI_LVL_SMALL_ROWS", small_rows)
    A, tv = problem("Stokes-C", 16)
    P = product_prec(A, tv, xml_params("Stokes-C", 16, 8, 1, partitioner="Skew Cartesian"), hostsim_lib)
    O = oracle_prec(A, tv, "Stokes-C", 16, 8, 1, partitioner="Skew Cartesian")
    b = np.random.default_rng(12).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-9


@pytest.mark.parametrize("levels,tol,re", [(0, 1e-9, 80.0), (1, 1e-8, 80.0), (1, 1e-8, 10000.0)])
def test_nonsymmetric_navier_stokes_like(hostsim_lib, levels, tol, re):
    """unsymmetric values (convection) on the Stokes pattern: the LU without pivoting and the separate L- and
    U-side panels against the oracle (SuperLU with partial pivoting).  re = 10000: cell Reynolds number 625, the
    velocity rows have lost diagonal dominance (diagonal / off-diagonal sum 0.14) -- robustness of the
    pivot-free factorisation (BASELINE configs[3] "robustness at high Re")."""
    from common import add_convection
    n = 16
    A, tv = problem("Stokes-C", n)
    A = add_convection(A, n, re=re)
    assert abs(A - A.T).max() > 1.0
    P = product_prec(A, tv, xml_params("Stokes-C", n, 8, levels, partitioner="Skew Cartesian"), hostsim_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, 8, levels, partitioner="Skew Cartesian")
    b = np.random.default_rng(21).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < tol


@pytest.mark.parametrize("levels", [0, 1])
def test_darcy3d_saddle_point(hostsim_lib, levels):
    """GaleriExt Darcy3D (BASELINE configs[4] in the small; Equations=Stokes-C as SURVEY 8d notes): velocities
    couple only through the pressures."""
    from oracle import galeri
    n = 16
    A = galeri.darcy3d(n, n, n, 1.0, -1.0)
    tv = galeri.create_testvector(A)
    P = product_prec(A, tv, xml_params("Stokes-C", n, 8, levels, partitioner="Skew Cartesian"), hostsim_lib)
    O = oracle_prec(A, tv, "Stokes-C", n, 8, levels, partitioner="Skew Cartesian")
    b = np.random.default_rng(22).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-8


def _cavity2d(re, grid=32):
    """the reference's own 2D driven-cavity linear system (testSuite/data/DrivenCavity/32x32/<re>, input of
    testSuite/cavity.xml), committed as a fixture by tests/golden/make_golden.py"""
    import os
    import scipy.sparse as sp
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drivencavity%d_2d_%s.npz" % (grid, re)))
    n = 3 * grid * grid
    A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, n))
    return A, z["rhs"], z["sol"]


def cavity2d_case(lib, re, part, levels, its_max, grid=32):
    from oracle import galeri, krylov
    A, rhs, sol = _cavity2d(re, grid)
    assert np.linalg.norm(A @ sol - rhs) <= 1e-12 * np.linalg.norm(rhs)          # the fixture is consistent
    tv = galeri.create_testvector(A)
    prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 2, "nx": grid, "ny": grid, "nz": 1},
           "Preconditioner": {"Separator Length": 4, "Number of Levels": levels, "Partitioner": part}}
    P = product_prec(A, tv, prm, lib)
    O = OraclePrec2D(A, tv, part, levels, grid)
    b = np.random.default_rng(31).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-9
    x, its, res = krylov.gmres(lambda v: A @ v, rhs, P.ApplyInverse, tol=1e-8, maxit=250)
    _, its_o, _ = krylov.gmres(lambda v: A @ v, rhs, O.apply_inverse, tol=1e-8, maxit=250)
    assert abs(its - its_o) <= 1 and its <= its_max
    vel = np.arange(A.shape[0]) % 3 != 2                                             # pressure is defined up to a constant
    assert np.linalg.norm((x - sol)[vel]) <= 1e-6 * np.linalg.norm(sol[vel])


def OraclePrec2D(A, tv, part, levels, grid=32):
    from oracle.hymls import Preconditioner as OraclePrec
    p = Params(nx=grid, ny=grid, nz=1, sx=4, levels=levels, equations="Stokes-C", dim=2, partitioner=part).finalize()
    return OraclePrec(A, p, testvector=tv).compute()


@pytest.mark.parametrize("re,part,levels,its_max", [
    ("re0", "Cartesian", 1, 60), ("re1000", "Cartesian", 1, 100), ("re1000", "Skew Cartesian", 1, 110), ("re1000", "Cartesian", 2, 120)])
def test_reference_driven_cavity_2d(hostsim_lib, re, part, levels, its_max):
    """2D Navier-Stokes Jacobians of the reference's test data (nonsymmetric at Re 1000), solved to the
    reference's stored solution; same GMRES iteration count as the oracle."""
    cavity2d_case(hostsim_lib, re, part, levels, its_max)


def test_reference_driven_cavity_2d_64(hostsim_lib):
    """the 64x64 Re 1000 system SURVEY 8d names as the reference-pinned robustness check (3-level)"""
    cavity2d_case(hostsim_lib, "re1000", "Cartesian", 2, 250, grid=64)


def test_tiled_separator_block_apply(hostsim_lib, monkeypatch):
    """large separator blocks are applied in 64-row tiles (one wave each); force the tiling on small blocks"""
    A, tv = problem("Stokes-C", 16)
    prm = xml_params("Stokes-C", 16, 8, 1, partitioner="Skew Cartesian")
    b = np.random.default_rng(14).uniform(-1, 1, A.shape[0])
    x_plain = product_prec(A, tv, prm, hostsim_lib).ApplyInverse(b)
    monkeypatch.setenv("HYMLS_MI_BLOCK_TILE_MIN", "8")
    x_tiled = product_prec(A, tv, prm, hostsim_lib).ApplyInverse(b)
    assert rel_diff(x_tiled, x_plain) < 1e-13


@pytest.mark.parametrize("eq,n,sx,levels,cx,part", [("Laplace", 16, 4, 2, 2, "Cartesian"), ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian")])
def test_multivector_apply_inverse(hostsim_lib, eq, n, sx, levels, cx, part):
    """ApplyInverse on an Epetra_MultiVector-like block (reference src/HYMLS_MatrixBlock.cpp:335-344: numvec columns):
    every column equals the single-vector result and the oracle's; strided host blocks and any nvec."""
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx=cx, partitioner=part), hostsim_lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx=cx, partitioner=part)
    rng = np.random.default_rng(8)
    for nvec in (2, 5, 9):
        B = rng.uniform(-1, 1, (A.shape[0], nvec))
        X = P.ApplyInverse(B)
        assert X.shape == B.shape
        for j in range(nvec):
            assert np.array_equal(X[:, j], P.ApplyInverse(B[:, j].copy()))
        assert rel_diff(X[:, nvec - 1], O.apply_inverse(B[:, nvec - 1])) < 1e-8


RETAIN = [
    ({"Retain Nodes": 2}, dict(rx=2)),
    ({"Retain Nodes (x)": 2, "Retain Nodes (z)": 3}, dict(retain_xyz=(2, -1, 3))),
    ({"Retain Nodes at Level 1": 2}, dict(retain_at_level={1: 2})),
    ({"Retain Nodes": 2, "Retain Nodes at Level 0": 1}, dict(rx=2, retain_at_level={0: 1})),
]


def retain_nodes_case(lib, extra, kw):
    """"Retain Nodes" > 1 (several V-sums per separator, Cartesian partitioner: reference
    src/HYMLS_CartesianPartitioner.cpp:289-296) with the reference's parameter precedence "(x|y|z)" > "at Level k" >
    "Retain Nodes" (src/HYMLS_BasePartitioner.cpp:108-137): level sizes and ApplyInverse equal the oracle's; more retained
    nodes never need more CG iterations."""
    from oracle import krylov
    A, tv = problem("Laplace", 16)
    P = product_prec(A, tv, xml_params("Laplace", 16, 4, 2, cx=2, extra=extra), lib)
    O = oracle_prec(A, tv, "Laplace", 16, 4, 2, cx=2, **kw)
    assert [s[1] for s in P.level_sizes()] == [s[1] for s in O.level_sizes()]
    b = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-10
    P1 = product_prec(A, tv, xml_params("Laplace", 16, 4, 2, cx=2), lib)
    assert P.level_sizes()[1][1] >= P1.level_sizes()[1][1]
    rhs = A @ np.random.default_rng(2).uniform(-1, 1, A.shape[0])
    its = krylov.pcg(lambda v: A @ v, rhs, P.ApplyInverse, tol=1e-8, maxit=200)[1]
    its1 = krylov.pcg(lambda v: A @ v, rhs, P1.ApplyInverse, tol=1e-8, maxit=200)[1]
    assert its <= its1


@pytest.mark.parametrize("extra,kw", RETAIN)
def test_retain_nodes(hostsim_lib, extra, kw):
    retain_nodes_case(hostsim_lib, extra, kw)


def test_retain_nodes_stokes_2d(hostsim_lib):
    """2D Stokes-C, Cartesian partitioner, two retained nodes per separator: oracle parity of the two-level method"""
    from oracle import galeri
    from oracle.hymls import Preconditioner as OraclePrec
    import scipy.sparse as sp
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drivencavity32_2d_re0.npz"))
    n = 32
    A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=(3 * n * n, 3 * n * n))
    tv = galeri.create_testvector(A)
    prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 2, "nx": n, "ny": n, "nz": 1},
           "Preconditioner": {"Separator Length": 8, "Number of Levels": 1, "Partitioner": "Cartesian", "Retain Nodes": 2}}
    P = product_prec(A, tv, prm, hostsim_lib)
    O = OraclePrec(A, Params(nx=n, ny=n, nz=1, sx=8, levels=1, equations="Stokes-C", dim=2, rx=2).finalize(), testvector=tv).compute()
    assert [s[1] for s in P.level_sizes()] == [s[1] for s in O.level_sizes()]
    b = np.random.default_rng(3).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-9


def test_separator_block_inversion_hostsim(hostsim_lib):
    """hymls_mi_invert_blocks through the C ABI on the simulator (the scalar Gauss-Jordan of the device interface)."""
    A, tv = problem("Laplace", 8)
    P = product_prec(A, tv, xml_params("Laplace", 8, 4, 0), hostsim_lib)
    rng = np.random.default_rng(3)
    B = rng.uniform(-1, 1, (3, 37, 37))
    B[1][np.arange(37), np.arange(37)] = 0.0
    X = P.InvertBlocks(B)
    for q in range(3):
        assert np.abs(X[q] @ B[q] - np.eye(37)).max() <= 1e-12 * np.linalg.cond(B[q])
    assert P.InvertBlocks(np.zeros((0, 5, 5))).shape == (0, 5, 5)


@pytest.mark.parametrize("eq,n,sx,levels,cx", [("Laplace", 16, 4, 1, -1), ("Stokes-C", 16, 8, 1, -1), ("Laplace", 16, 4, 2, 2)])
def test_recompute_keeps_the_coarse_plan_hostsim(hostsim_lib, eq, n, sx, levels, cx):
    """SetMatrix + Compute with the same pattern: the coarse direct solver keeps its ordering and symbolic factorisation and
    only refactors (CoarseSolver::Compute keeps its Amesos solver in the reference, src/HYMLS_CoarseSolver.cpp:131-152);
    the result must be what a fresh preconditioner for the new matrix gives, also after going back to the first matrix."""
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), hostsim_lib)
    b = np.random.default_rng(11).uniform(-1, 1, A.shape[0])
    rng = np.random.default_rng(12)
    A2 = A.copy()
    A2.data = A2.data * (1.0 + 0.05 * rng.uniform(-1, 1, A2.data.size))       # same pattern, other values
    if eq == "Stokes-C":
        A2 = A.copy(); A2.data = A2.data * 1.5
    P.SetMatrix(A2)
    P.Compute()
    F = product_prec(A2, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), hostsim_lib)
    assert rel_diff(P.ApplyInverse(b), F.ApplyInverse(b)) < 1e-12
    # back to the first matrix: again a refactorisation
    P.SetMatrix(A)
    P.Compute()
    F = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, "Skew Cartesian" if eq == "Stokes-C" else "Cartesian"), hostsim_lib)
    assert rel_diff(P.ApplyInverse(b), F.ApplyInverse(b)) < 1e-12
    assert P.NumInitialize() == 1 and P.NumCompute() == 3


CLASS_CASES = [
    # eq, n, sx, levels, cx, partitioner, extra
    ("Laplace", 24, 4, 2, 2, "Cartesian", None),
    ("Stokes-C", 16, 4, 0, -1, "Cartesian", None),
    ("Stokes-C", 32, 8, 1, -1, "Skew Cartesian", None),
    ("Stokes-C", 24, 4, 2, 2, "Skew Cartesian", None),
    ("Stokes-C", 8, 4, 0, -1, "Skew Cartesian", {"x-periodic": True, "y-periodic": True, "z-periodic": True}),
]


@pytest.mark.parametrize("eq,n,sx,levels,cx,part,extra", CLASS_CASES)
def test_pattern_class_shortcut_agrees_with_the_full_patterns(hostsim_lib, monkeypatch, capfd, eq, n, sx, levels, cx, part, extra):
    """Initialize classifies a subdomain by a signature (node list relative to its first node, hashes of the matrix rows
    relative to their row number, the subdomains listing every separator node, group structure, relative coordinates) and
    builds the extended pattern only for the first subdomain of every signature (csrc/precond.cpp, build_classes).
    HYMLS_MI_VERIFY_CLASSES builds every pattern nevertheless and fails with -3 if the shortcut would have chosen another
    class; the results with the shortcut, with the verification and without the shortcut are the same bits."""
    import hymls_amd
    if eq == "Laplace":
        K = hymls_amd.generate_problem("Laplace", n, n, n, lib=hostsim_lib)
    else:
        K = hymls_amd.generate_problem("Stokes", n, n, n, lib=hostsim_lib, periodic=(True, True, True) if extra else (False, False, False))
    tv = hymls_amd.generate_testvector(*K, lib=hostsim_lib)
    prm = xml_params(eq, n, sx, levels, cx, part)
    if extra:
        prm["Problem"].update(extra)
    b = np.random.default_rng(5).uniform(-1, 1, K[0].size - 1)
    out = []
    for env in ({}, {"HYMLS_MI_VERIFY_CLASSES": "1"}, {"HYMLS_MI_NO_FAST_CLASSES": "1"}):
        for k in ("HYMLS_MI_VERIFY_CLASSES", "HYMLS_MI_NO_FAST_CLASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("HYMLS_MI_PATTERN_PROF", "1")
        P = hymls_amd.Preconditioner(K, prm, testVector=tv, lib=hostsim_lib)
        assert P.Initialize() == 0
        if extra:
            # (periodic Stokes is singular without a border, tests/test_periodic.py: here only the classification runs)
            x = np.asarray(P.level_sizes(), dtype=np.float64).ravel()
        else:
            assert P.Compute() == 0
            x = P.ApplyInverse(b)
        out.append(x)
        err = capfd.readouterr().err
        if not env:
            import re
            m = re.search(r"pattern classes: (\d+) of (\d+) subdomains classified by signature", err)
            assert m, err[-500:]
            if int(m.group(2)) >= 64:
                assert int(m.group(1)) > 0          # the shortcut is actually taken where translates exist
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
