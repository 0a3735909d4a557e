"""Test-matrix generators (oracle side): restatement of the reference's GaleriExt
generators for the C grid, with and without periodic directions.  TEST INFRASTRUCTURE ONLY.

Reference:
  src/GaleriExt_Stokes3D.h:89-285   (Stokes3D, grid_type 'C')
  src/GaleriExt_Darcy3D.h:45-176    (Darcy3D)
  src/GaleriExt_Periodic.cpp:36-66  (GetNeighboursCartesian3d; -1 outside box)
  src/HYMLS_MainUtils.cpp:260-348   (create_matrix: Stokes a=nx^2, b=1; Laplace scaled by -1)
  Trilinos Galeri Cross3D semantics (not in /root/reference):
  diag a, neighbours b..g, missing neighbours omitted, cell id=(k*ny+j)*nx+i.

GID convention: gid = cell*dof + var (src/HYMLS_Tools.cpp:691-723).
"""
import numpy as np
import scipy.sparse as sp


def _cells(nx, ny, nz):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    return i.ravel(), j.ravel(), k.ravel()


def _neigh(nx, ny, nz, perio=(False, False, False)):
    """left,right,lower,upper,below,above cell ids (-1 outside; wrapped in periodic directions),
    GaleriExt_Periodic.cpp:8-66."""
    i, j, k = _cells(nx, ny, nz)
    c = (k * ny + j) * nx + i

    def cell(ii, jj, kk):
        return (kk * ny + jj) * nx + ii
    if perio[0]:
        left, right = cell((i - 1) % nx, j, k), cell((i + 1) % nx, j, k)
    else:
        left, right = np.where(i > 0, c - 1, -1), np.where(i < nx - 1, c + 1, -1)
    if perio[1]:
        lower, upper = cell(i, (j - 1) % ny, k), cell(i, (j + 1) % ny, k)
    else:
        lower, upper = np.where(j > 0, c - nx, -1), np.where(j < ny - 1, c + nx, -1)
    if perio[2]:
        below, above = cell(i, j, (k - 1) % nz), cell(i, j, (k + 1) % nz)
    else:
        below, above = np.where(k > 0, c - nx * ny, -1), np.where(k < nz - 1, c + nx * ny, -1)
    return c, (left, right, lower, upper, below, above)


def laplace3d(nx, ny, nz, scale=-1.0):
    """Galeri 'Laplace3D' = Cross3D(6,-1,...), scaled by -1 as in create_matrix
    (src/HYMLS_MainUtils.cpp:340-345).  dof = 1."""
    c, nb = _neigh(nx, ny, nz)
    rows = [c]
    cols = [c]
    vals = [np.full(c.size, 6.0)]
    for n in nb:
        m = n >= 0
        rows.append(c[m])
        cols.append(n[m])
        vals.append(np.full(m.sum(), -1.0))
    A = sp.coo_matrix((np.concatenate(vals) * scale, (np.concatenate(rows), np.concatenate(cols))),
                      shape=(c.size, c.size)).tocsr()
    A.sort_indices()
    return A


def darcy3d(nx, ny, nz, a, b, perio=(False, False, False)):
    """GaleriExt::Darcy3D (src/GaleriExt_Darcy3D.h:45-176): A=diag(a) on u,v,w,
    grad entries (-b at p_here, +b at p_next), div rows with c=-b; neighbours wrap in periodic directions."""
    dof = 4
    c, (left, right, lower, upper, below, above) = _neigh(nx, ny, nz, perio)
    N = c.size * dof
    rows, cols, vals = [], [], []
    cc = -b

    def add(r, cl, v):
        rows.append(r)
        cols.append(cl)
        vals.append(np.full(r.size, v) if np.isscalar(v) else v)

    for var, nxt in ((0, right), (1, upper), (2, above)):
        add(c * dof + var, c * dof + var, a)
        m = nxt >= 0
        add(c[m] * dof + var, c[m] * dof + 3, -b)
        add(c[m] * dof + var, nxt[m] * dof + 3, b)
    for var, nxt, prv in ((0, right, left), (1, upper, lower), (2, above, below)):
        m = nxt >= 0
        add(c[m] * dof + 3, c[m] * dof + var, -cc)
        m = prv >= 0
        add(c[m] * dof + 3, prv[m] * dof + var, cc)
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(N, N)).tocsr()
    A.sort_indices()
    return A


def stokes3d(nx, ny, nz, a=None, b=1.0, perio=(False, False, False)):
    """GaleriExt::Stokes3D, C grid (src/GaleriExt_Stokes3D.h:89-285).
    Default a = nx*nx, b = 1 as in create_matrix (src/HYMLS_MainUtils.cpp:322-323).
    Explicit zeros written by the reference (removed wall couplings) are dropped
    here: the fixture test compares by matvec, as the reference's own test does.
    With any periodic direction (restated as written, quirk included): the gradient / divergence part and every wall
    decision use the PERIODIC neighbours, but the velocity Laplacians are GaleriExt Cross3DN matrices
    (get3DLaplaceMatrixForVar :77-80, src/GaleriExt_Cross3DN.h:55-134), which know nothing of the periodicity: the
    couplings across a periodic boundary are absent and every missing neighbour's -1 is added to the diagonal
    (homogeneous Neumann).  No reference fixture holds a periodic matrix: parity of the periodic VALUES is unpinned;
    the integration target stokes4_3D.xml pins what is built on them."""
    if a is None:
        a = float(nx * nx)
    dof = 4
    any_perio = any(perio)
    c, (left, right, lower, upper, below, above) = _neigh(nx, ny, nz, perio)
    _, lap_nb = _neigh(nx, ny, nz)            # the Laplacians' own (non-periodic) neighbours
    ncell = c.size
    N = ncell * dof
    rows, cols, vals = [], [], []

    def add(r, cl, v):
        r = np.asarray(r)
        rows.append(r)
        cols.append(np.asarray(cl))
        vals.append(np.full(r.size, float(v)) if np.isscalar(v) else np.asarray(v, dtype=float))

    def nxt_of(arr, n):
        out = np.full(ncell, -1)
        m = n >= 0
        out[m] = arr[n[m]]
        return out

    # Darcy3D(map, 0.0, -b): grad entries +b at p_here, -b at p_next; div rows c = b
    for var, nxt, prv in ((0, right, left), (1, upper, lower), (2, above, below)):
        m = nxt >= 0
        add(c[m] * dof + var, c[m] * dof + 3, b)
        add(c[m] * dof + var, nxt[m] * dof + 3, -b)
        add(c[m] * dof + 3, c[m] * dof + var, -b)
        m = prv >= 0
        add(c[m] * dof + 3, prv[m] * dof + var, b)

    # velocity rows: -a * Cross3D(6,-1) with wall corrections (Stokes3D.h:150-262)
    tang = {0: ((lower, upper), (below, above)),
            1: ((left, right), (below, above)),
            2: ((left, right), (lower, upper))}
    nrm = {0: right, 1: upper, 2: above}
    lap_diag = np.full(ncell, 6.0)
    if any_perio:                              # Cross3DN: diag = a + sum of the missing neighbours' coefficients
        for n in lap_nb:
            lap_diag -= (n < 0)
    for var in range(3):
        nx_ = nrm[var]
        nxnx = nxt_of(nx_, nx_)  # "rightright" etc.
        wall = nx_ < 0  # velocity sits on the wall: Dirichlet row, diag = 1
        add(c[wall] * dof + var, c[wall] * dof + var, 1.0)
        inner = ~wall
        add_diag = np.zeros(ncell)
        for (lo, hi) in tang[var]:
            add_diag += np.where((lo < 0) | (hi < 0), a, 0.0)
        add(c[inner] * dof + var, c[inner] * dof + var, -(lap_diag[inner] * a + add_diag[inner]))
        for ni, n in enumerate(lap_nb):
            m = inner & (n >= 0)
            # coupling to the velocity on the wall is removed (Stokes3D.h:199-206)
            if ni == 2 * var + 1:
                m = m & ~((nx_ >= 0) & (nxnx < 0))
            add(c[m] * dof + var, n[m] * dof + var, a)
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(N, N)).tocsr()
    A.sort_indices()
    return A


def oseen3d(nx, ny, nz, re, a=None, b=1.0):
    """BASELINE configs[3]: the reference ships no 3D Navier-Stokes Jacobian at Re > 0
    (testSuite/cavity3D.xml reads data/DrivenCavity/32x32x32/Re0/jac.mtx, a missing blob), so the input is
    synthesised: Stokes3D(a, b) (velocity rows scaled like the reference's, i.e. multiplied by Re) plus the
    central difference of (w . grad) u on every existing velocity-velocity coupling: +g w_d to the next,
    -g w_d to the previous neighbour in direction d, g = a Re / (2 nx) (cell Peclet number Re |w| h / 2),
    w = (-y + 0.3 z, x - 0.2 z, 0.5 x y) at the cell centre, x = (i + 1/2)/nx - 1/2 etc.  Gradient and
    divergence entries are untouched (F-matrix), the pattern is that of Stokes3D.  Parity unpinned by the
    reference (no fixture exists); the product generator is pinned to this one bit for bit."""
    if a is None:
        a = float(nx * nx)
    A = stokes3d(nx, ny, nz, a, b).tocsr()
    A.sort_indices()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    cols = A.indices
    var_r, var_c = rows % 4, cols % 4
    cell_r, cell_c = rows // 4, cols // 4
    diff = cell_c - cell_r
    i, j, k = cell_r % nx, (cell_r // nx) % ny, cell_r // (nx * ny)
    x, y, z = (i + 0.5) / nx - 0.5, (j + 0.5) / ny - 0.5, (k + 0.5) / nz - 0.5
    w = [-y + 0.3 * z, x - 0.2 * z, (0.5 * x) * y]
    g = re / (2.0 * nx) * a
    nent = np.diff(A.indptr)[rows]
    data = A.data.copy()
    for d, st in enumerate((1, nx, nx * ny)):
        for sgn in (+1, -1):
            m = (var_r < 3) & (var_r == var_c) & (diff == sgn * st) & (nent > 1)
            data[m] += (sgn * g) * w[d][m]
    A.data = data
    return A


def create_testvector(A):
    """create_testvector (src/HYMLS_MainUtils.cpp:208-258), Stokes-C / Laplace:
    all ones, zero on rows whose only nonzero is the diagonal."""
    A = A.tocsr()
    t = np.ones(A.shape[0])
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    offd = (A.indices != rows) & (A.data != 0.0)
    has_off = np.zeros(A.shape[0], dtype=bool)
    has_off[rows[offd]] = True
    t[~has_off] = 0.0
    return t
