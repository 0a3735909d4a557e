"""Bordered systems [K V; W' C] (HYMLS::BorderedOperator interface of the preconditioner).
The exactness tests restate the reference's unit tests BorderedApplyInverse / BorderedApplyInverse_without_C
(testSuite/unit_tests/HYMLS_Preconditioner.cpp:278-378: random V, W (n x 2), random or zero C, one-level
preconditioner = exact bordered solve, 1e-10); the multi-level cases compare with the oracle."""
import numpy as np
import pytest

import hymls_amd
from common import problem, xml_params, rel_diff
from oracle.partition import Params
from oracle.hymls import Preconditioner as OraclePrec


def bordered_exact(lib, with_c):
    A, tv = problem("Laplace", 8)
    N, m = A.shape[0], 2
    rng = np.random.default_rng(41)
    V, W = rng.uniform(-1, 1, (N, m)), rng.uniform(-1, 1, (N, m))
    C = rng.uniform(-1, 1, (m, m)) if with_c else None
    P = hymls_amd.Preconditioner(A, xml_params("Laplace", 8, 4, 0), testVector=tv, lib=lib)
    assert P.Initialize() == 0
    assert P.SetBorder(V, W, C) == 0 and P.HaveBorder() and not P.IsComputed()
    assert P.Compute() == 0
    x_ex, s_ex = rng.uniform(-1, 1, N), (rng.uniform(-1, 1, m) if with_c else np.zeros(m))
    Cm = C if with_c else np.zeros((m, m))
    b = A @ x_ex + V @ s_ex
    t = W.T @ x_ex + Cm @ s_ex
    x, s = P.ApplyInverseBordered(b, t)
    assert np.abs(x - x_ex).max() < 1e-10 and np.abs(s - s_ex).max() < 1e-10
    # plain ApplyInverse of a bordered operator: border right-hand side 0, S dropped
    x0, _ = P.ApplyInverseBordered(b, np.zeros(m))
    assert np.array_equal(P.ApplyInverse(b), x0)
    # removing the border gives the plain preconditioner back
    P.SetBorder(None)
    P.Compute()
    assert np.abs(P.ApplyInverse(A @ x_ex) - x_ex).max() < 1e-10


CASES = [("Laplace", 16, 4, 1, -1, "Cartesian"), ("Stokes-C", 8, 4, 0, -1, "Skew Cartesian"),
         ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian"), ("Laplace", 16, 4, 2, 2, "Cartesian"),
         ("Stokes-C", 16, 4, 2, 2, "Skew Cartesian")]


def bordered_vs_oracle(lib, eq, n, sx, levels, cx, part, v_is_w=False):
    A, tv = problem(eq, n)
    N, m = A.shape[0], 2
    rng = np.random.default_rng(42)
    V = rng.uniform(-1, 1, (N, m))
    W = V if v_is_w else rng.uniform(-1, 1, (N, m))
    C = rng.uniform(-1, 1, (m, m))
    O = OraclePrec(A, Params(nx=n, ny=n, nz=n, sx=sx, levels=levels, equations=eq, partitioner=part, cx=cx).finalize(), testvector=tv)
    O.set_border(V, W, C)
    O.compute()
    P = hymls_amd.Preconditioner(A, xml_params(eq, n, sx, levels, cx, part), testVector=tv, lib=lib)
    P.Initialize()
    P.SetBorder(V, None if v_is_w else W, C)
    P.Compute()
    b, t = rng.uniform(-1, 1, N), rng.uniform(-1, 1, m)
    xo, so = O.apply_inverse_bordered(b, t)
    xp, sp = P.ApplyInverseBordered(b, t)
    assert rel_diff(xp, xo) < 1e-9 and rel_diff(sp, so) < 1e-8
    # SetMatrix with the same pattern + Compute keeps the border (reference: computed_ = false, border stays)
    P.SetMatrix(A * 2.0)
    P.Compute()
    O2 = OraclePrec(A * 2.0, O.params, testvector=tv)
    O2.set_border(V, W, C)
    O2.compute()
    xo2, so2 = O2.apply_inverse_bordered(b, t)
    xp2, sp2 = P.ApplyInverseBordered(b, t)
    assert rel_diff(xp2, xo2) < 1e-9 and rel_diff(sp2, so2) < 1e-8


@pytest.mark.parametrize("with_c", [True, False])
def test_bordered_exact_hostsim(hostsim_lib, with_c):
    bordered_exact(hostsim_lib, with_c)


@pytest.mark.parametrize("eq,n,sx,levels,cx,part", CASES)
def test_bordered_vs_oracle_hostsim(hostsim_lib, eq, n, sx, levels, cx, part):
    bordered_vs_oracle(hostsim_lib, eq, n, sx, levels, cx, part)


def test_bordered_w_defaults_to_v_hostsim(hostsim_lib):
    bordered_vs_oracle(hostsim_lib, "Stokes-C", 16, 8, 1, -1, "Skew Cartesian", v_is_w=True)


@pytest.mark.gpu
@pytest.mark.parametrize("with_c", [True, False])
def test_bordered_exact_gpu(gpu_lib, with_c):
    bordered_exact(gpu_lib, with_c)


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,sx,levels,cx,part", CASES)
def test_bordered_vs_oracle_gpu(gpu_lib, eq, n, sx, levels, cx, part):
    bordered_vs_oracle(gpu_lib, eq, n, sx, levels, cx, part)


def bordered_krylov(lib, dev):
    """BorderedSolver: GMRES on the augmented vectors with the bordered multi-level preconditioner converges to the
    solution of [K V; W' C] [x; s] = [b; t] in as many iterations as the same loop with the oracle's operator."""
    import torch
    from oracle import krylov
    eq, n, sx, levels, part = "Stokes-C", 16, 8, 1, "Skew Cartesian"
    A, tv = problem(eq, n)
    N, m = A.shape[0], 2
    rng = np.random.default_rng(43)
    V, W, C = rng.uniform(-1, 1, (N, m)), rng.uniform(-1, 1, (N, m)), rng.uniform(-1, 1, (m, m))
    P = hymls_amd.Preconditioner(A, xml_params(eq, n, sx, levels, -1, part), testVector=tv, lib=lib)
    P.Initialize()
    S = hymls_amd.BorderedSolver(P, P, {"Krylov Method": "GMRES", "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 300, "Num Blocks": 300}})
    S.SetBorder(V, W, C, device=dev)
    P.Compute()
    x_ex, s_ex = rng.uniform(-1, 1, N), rng.uniform(-1, 1, m)
    b, t = A @ x_ex + V @ s_ex, W.T @ x_ex + C @ s_ex
    x, s = S.ApplyInverse(torch.from_numpy(b).to(dev), t)
    x = x.cpu().numpy()
    assert np.linalg.norm(A @ x + V @ s - b) <= 1e-7 * np.linalg.norm(b) and np.abs(W.T @ x + C @ s - t).max() < 1e-6
    O = OraclePrec(A, Params(nx=n, ny=n, nz=n, sx=sx, levels=levels, equations=eq, partitioner=part).finalize(), testvector=tv)
    O.set_border(V, W, C)
    O.compute()

    def op(z):
        return np.concatenate([A @ z[:N] + V @ z[N:], W.T @ z[:N] + C @ z[N:]])

    def pr(z):
        xx, ss = O.apply_inverse_bordered(z[:N], z[N:])
        return np.concatenate([xx, ss])

    _, its_o, _ = krylov.gmres(op, np.concatenate([b, t]), pr, tol=1e-8, maxit=300)
    assert abs(S.getNumIter() - its_o) <= 1


def test_bordered_krylov_hostsim(hostsim_lib):
    bordered_krylov(hostsim_lib, "cpu")


@pytest.mark.gpu
def test_bordered_krylov_gpu(gpu_lib):
    bordered_krylov(gpu_lib, "cuda")


def singular_with_border(lib, levels, cx):
    """Stokes without "Fix Pressure Level": K is singular (constant pressure mode), the border [K v; v' 0] with v the
    constant-pressure vector makes the system regular (what testSuite/cavity.xml asks for with "Null Space Type" =
    "Constant P").  The oracle factors the AugmentedMatrix of the coarsest level with pivoting (as the reference's KLU);
    the product moves one pressure node of the coarse matrix into the border instead."""
    from dataclasses import replace
    eq, n, sx, part = "Stokes-C", 16, 4 if levels == 2 else 8, "Skew Cartesian"
    A, tv = problem(eq, n)
    N = A.shape[0]
    v = np.zeros((N, 1)); v[3::4, 0] = 1.0
    assert np.abs(A @ v).max() < 1e-9                     # v spans the null space
    op = Params(nx=n, ny=n, nz=n, sx=sx, levels=levels, equations=eq, partitioner=part, cx=cx).finalize()
    op = replace(op, fix_gids=[])
    O = OraclePrec(A, op, testvector=tv)
    O.set_border(v, v, np.zeros((1, 1)))
    O.compute()
    prm = xml_params(eq, n, sx, levels, cx, part, extra={"Fix Pressure Level": False})
    P = hymls_amd.Preconditioner(A, prm, testVector=tv, lib=lib)
    P.Initialize()
    P.SetBorder(v)
    P.Compute()
    rng = np.random.default_rng(44)
    b, t = rng.uniform(-1, 1, N), rng.uniform(-1, 1, 1)
    xo, so = O.apply_inverse_bordered(b, t)
    xp, sp = P.ApplyInverseBordered(b, t)
    assert np.isfinite(xp).all()
    assert rel_diff(xp, xo) < 1e-8 and abs(sp[0] - so[0]) <= 1e-8 * max(1.0, abs(so[0]))


@pytest.mark.parametrize("levels,cx", [(0, -1), (1, -1), (2, 2)])
def test_singular_matrix_regular_bordered_hostsim(hostsim_lib, levels, cx):
    singular_with_border(hostsim_lib, levels, cx)


@pytest.mark.gpu
@pytest.mark.parametrize("levels,cx", [(1, -1), (2, 2)])
def test_singular_matrix_regular_bordered_gpu(gpu_lib, levels, cx):
    singular_with_border(gpu_lib, levels, cx)
