"""The compiled CPU oracle (oracle/cpu/hymls_cpu.cpp + oracle/cpu_oracle.py) is pinned by the numpy oracle
(oracle/hymls.py, itself pinned by the reference's fixtures: tests/test_oracle_pins.py): same ApplyInverse to rounding on
every configuration family, same Krylov iteration counts on the reference's own 16^3 Stokes system.  With that it may
serve as the checker at sizes the numpy oracle cannot reach in test time (tests/test_gpu_parity.py) and as bench.py's
CPU baseline."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import galeri, krylov, cpu_oracle
from oracle.partition import Params
from oracle.hymls import Preconditioner as NumpyOracle
from common import rel_diff, add_convection

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [
    # eq, n, sx, levels, cx, partitioner, matrix
    ("Laplace", 16, 4, 0, -1, "Cartesian", "laplace"),
    ("Laplace", 16, 4, 2, 2, "Cartesian", "laplace"),
    ("Stokes-C", 16, 8, 0, -1, "Skew Cartesian", "stokes"),
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", "stokes"),
    ("Stokes-C", 16, 4, 2, 2, "Skew Cartesian", "stokes"),
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", "darcy"),
    ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian", "oseen"),
    ("Stokes-C", 12, 4, 1, -1, "Skew Cartesian", "stokes"),      # ragged: 12 is not a multiple of 8 = sx * cx
]


def matrix(kind, n):
    return {"laplace": lambda: galeri.laplace3d(n, n, n), "stokes": lambda: galeri.stokes3d(n, n, n),
            "darcy": lambda: galeri.darcy3d(n, n, n, 1.0, -1.0), "oseen": lambda: galeri.oseen3d(n, n, n, 125.0)}[kind]()


@pytest.mark.parametrize("eq,n,sx,levels,cx,part,kind", CASES)
def test_compiled_oracle_matches_numpy_oracle(eq, n, sx, levels, cx, part, kind):
    A = matrix(kind, n)
    tv = galeri.create_testvector(A)
    p = Params(nx=n, ny=n, nz=n, sx=sx, cx=cx, levels=levels, equations=eq, partitioner=part).finalize()
    O = NumpyOracle(A, p, testvector=tv).compute()
    Cq = cpu_oracle.Preconditioner(A, p, testvector=tv, nthreads=4).compute()
    assert Cq.flags == 0 and Cq.level_sizes() == O.level_sizes()
    rng = np.random.default_rng(3)
    for _ in range(2):
        b = rng.uniform(-1, 1, A.shape[0])
        assert rel_diff(Cq.apply_inverse(b), O.apply_inverse(b)) < 1e-9
    Cq.set_threads(1)                     # the thread count does not change a bit
    x4 = Cq.apply_inverse(b); Cq.set_threads(3)
    assert np.array_equal(Cq.apply_inverse(b), x4)


def test_compiled_oracle_reaches_the_reference_iteration_targets():
    """reference integration test stokes2_3D.xml (16^3 fixture, Skew sx=4, cx=2, XML levels 2, GMRES 1e-8: <= 145
    iterations, relative error 1e-5) through the compiled oracle: same iteration count as the numpy oracle."""
    n = 16
    A = galeri.stokes3d(n, n, n)
    tv = galeri.create_testvector(A)
    p = Params(nx=n, ny=n, nz=n, sx=4, cx=2, levels=2, equations="Stokes-C", partitioner="Skew Cartesian").finalize()
    O = NumpyOracle(A, p, testvector=tv).compute()
    Cq = cpu_oracle.Preconditioner(A, p, testvector=tv, nthreads=4).compute()
    x_ex = np.random.default_rng(11).uniform(-1, 1, A.shape[0])
    rhs = A @ x_ex
    _, its_o, _ = krylov.gmres(lambda v: A @ v, rhs, O.apply_inverse, tol=1e-8, maxit=250)
    xc, its_c, res = krylov.gmres(lambda v: A @ v, rhs, Cq.apply_inverse, tol=1e-8, maxit=250)
    assert abs(its_c - its_o) <= 1 and its_c <= 145 and res < 1e-7
    vel = np.arange(A.shape[0]) % 4 != 3
    assert np.linalg.norm((xc - x_ex)[vel]) <= 1e-5 * np.linalg.norm(x_ex[vel])


def test_cartesian_zero_pressure_blocks_are_flagged():
    """the Cartesian / 3D Stokes-C finding (tests/test_oracle_pins.py) seen from the compiled oracle: dgetrf INFO > 0"""
    n = 8
    A = galeri.stokes3d(n, n, n)
    p = Params(nx=n, ny=n, nz=n, sx=4, levels=1, equations="Stokes-C").finalize()
    Cq = cpu_oracle.Preconditioner(A, p, testvector=galeri.create_testvector(A)).compute()
    assert Cq.flags & 2
