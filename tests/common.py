"""Shared helpers of the parity tests: build the same problem for the product (through the
C ABI) and for the oracle, on the same seeded inputs."""
import numpy as np

from oracle import galeri
from oracle.partition import Params
from oracle.hymls import Preconditioner as OraclePrec
import hymls_amd


def problem(eq, n, nz=None):
    nz = n if nz is None else nz
    if eq == "Laplace":
        A = galeri.laplace3d(n, n, nz)
    else:
        A = galeri.stokes3d(n, n, nz)
    return A, galeri.create_testvector(A)


def xml_params(eq, n, sx, levels, cx=-1, partitioner="Cartesian", nz=None, extra=None):
    nz = n if nz is None else nz
    prec = {"Separator Length": sx, "Number of Levels": levels, "Partitioner": partitioner}
    if cx > 0:
        prec["Coarsening Factor"] = cx
    if extra:
        prec.update(extra)
    return {"Problem": {"Equations": eq, "Dimension": 3, "nx": n, "ny": n, "nz": nz}, "Preconditioner": prec}


def oracle_prec(A, tv, eq, n, sx, levels, cx=-1, nz=None, **kw):
    nz = n if nz is None else nz
    p = Params(nx=n, ny=n, nz=nz, sx=sx, cx=cx, levels=levels, equations=eq, **kw).finalize()
    return OraclePrec(A, p, testvector=tv).compute()


def product_prec(A, tv, prm, lib):
    P = hymls_amd.Preconditioner(A, prm, testVector=tv, lib=lib)
    assert not P.IsInitialized() and not P.IsComputed()
    assert P.Initialize() == 0 and P.IsInitialized()
    assert P.Compute() == 0 and P.IsComputed()
    return P


def rel_diff(x, y):
    return np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)


def add_convection(A, n, re=50.0, nz=None):
    """Stokes3D Jacobian + a linearised convection term (BASELINE configs[3] in the small: a Navier-Stokes-like,
    NONSYMMETRIC F-matrix): for every velocity-velocity coupling to the neighbour in direction d the central
    difference of w_d * du/dx_d is added (+g to the next, -g to the previous neighbour), with w a fixed swirling
    field and g = re / (2 n) * a / n (cell Reynolds number re / n relative to the diffusion coefficient).
    Gradient and divergence blocks are untouched, the sparsity pattern does not change."""
    nz = n if nz is None else nz
    A = A.tocsr().copy()
    A.sort_indices()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    cols = A.indices
    var_r, var_c = rows % 4, cols % 4
    cell_r, cell_c = rows // 4, cols // 4
    step = {1: 0, n: 1, n * n: 2}
    diff = cell_c - cell_r
    i, j, k = cell_r % n, (cell_r // n) % n, cell_r // (n * n)
    x, y, z = (i + 0.5) / n - 0.5, (j + 0.5) / n - 0.5, (k + 0.5) / nz - 0.5
    w = np.stack([-y + 0.3 * z, x - 0.2 * z, 0.5 * x * y])       # swirling, not divergence-free on purpose
    a_diff = float(n * n)
    g = re / (2.0 * n) * a_diff / n
    data = A.data.copy()
    for st, d in step.items():
        for sgn in (+1, -1):
            m = (var_r < 3) & (var_r == var_c) & (diff == sgn * st) & (A.data != 0)
            data[m] += sgn * g * w[d][m]
    A.data = data
    return A
