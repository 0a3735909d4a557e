"""hymls_amd.Solver (the reference's BaseSolver semantics on device tensors).  -m "not gpu": CPU torch tensors
through the TEST-ONLY host simulator; the iteration counts are the reference's integration-test targets and
must equal the oracle's."""
import numpy as np
import pytest
import torch

import hymls_amd
from common import problem, xml_params, oracle_prec, product_prec
from oracle import krylov


def _solve(lib, dev, eq, n, sx, levels, cx, part, solver_params, seed):
    A, tv = problem(eq, n)
    P = product_prec(A, tv, xml_params(eq, n, sx, levels, cx, part), lib)
    O = oracle_prec(A, tv, eq, n, sx, levels, cx, partitioner=part)
    rng = np.random.default_rng(seed)
    x_ex = rng.uniform(-1, 1, A.shape[0]); b = A @ x_ex
    S = hymls_amd.Solver(P, P, {"Solver": solver_params})
    x = S.ApplyInverse(torch.from_numpy(b).to(dev))
    return A, O, b, x.cpu().numpy(), S


def check_cg(lib, dev):
    # threeD1.xml: Laplace 32^3, sx=4, 3-level, CG 1e-10: reference target <= 35 iterations
    A, O, b, x, S = _solve(lib, dev, "Laplace", 32, 4, 2, -1, "Cartesian",
                           {"Krylov Method": "CG", "Iterative Solver": {"Convergence Tolerance": 1e-10, "Maximum Iterations": 100}}, 3)
    _, its_o, _ = krylov.pcg(lambda v: A @ v, b, O.apply_inverse, tol=1e-10, maxit=100)
    assert abs(S.getNumIter() - its_o) <= 1 and S.getNumIter() <= 35
    assert np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 1e-9


def check_gmres(lib, dev):
    # stokes1_3D.xml shape at 16^3: right-preconditioned GMRES, 1e-8; with a short restart length as well
    prm = {"Krylov Method": "GMRES", "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 400, "Num Blocks": 250}}
    A, O, b, x, S = _solve(lib, dev, "Stokes-C", 16, 8, 1, -1, "Skew Cartesian", prm, 6)
    _, its_o, _ = krylov.gmres(lambda v: A @ v, b, O.apply_inverse, tol=1e-8, maxit=250)
    assert abs(S.getNumIter() - its_o) <= 1 and S.getNumIter() <= 130
    assert np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 1e-7
    prm["Iterative Solver"]["Num Blocks"] = 30
    A, O, b, x, S2 = _solve(lib, dev, "Stokes-C", 16, 8, 1, -1, "Skew Cartesian", prm, 6)
    assert S2.getNumIter() >= S.getNumIter() and np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 1e-7
    prm["Left or Right Preconditioning"] = "Left"
    prm["Iterative Solver"]["Num Blocks"] = 250
    A, O, b, x, S3 = _solve(lib, dev, "Stokes-C", 16, 8, 1, -1, "Skew Cartesian", prm, 6)
    assert S3.getNumIter() <= 140


def test_solver_cg_hostsim(hostsim_lib):
    check_cg(hostsim_lib, "cpu")


def test_solver_gmres_hostsim(hostsim_lib):
    check_gmres(hostsim_lib, "cpu")


def test_solver_not_converged_raises(hostsim_lib):
    A, tv = problem("Laplace", 16)
    P = product_prec(A, tv, xml_params("Laplace", 16, 4, 1), hostsim_lib)
    S = hymls_amd.Solver(P, P, {"Krylov Method": "CG", "Iterative Solver": {"Convergence Tolerance": 1e-12, "Maximum Iterations": 2}})
    with pytest.raises(RuntimeError):
        S.ApplyInverse(torch.ones(A.shape[0], dtype=torch.float64))
    assert S.getNumIter() == 2


@pytest.mark.gpu
def test_solver_cg_gpu(gpu_lib):
    check_cg(gpu_lib, "cuda")


@pytest.mark.gpu
def test_solver_gmres_gpu(gpu_lib):
    check_gmres(gpu_lib, "cuda")
