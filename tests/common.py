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
