"""ctypes binding of include/hymls_mi.h and the reference-shaped Python operator.

``Preconditioner`` keeps the reference's method names and lifecycle
(SetParameters -> Initialize -> Compute -> ApplyInverse*, reference
src/HYMLS_Preconditioner.hpp:93-127): ``Compute`` auto-initialises, ``ApplyInverse``
before ``Compute`` is an error (src/HYMLS_Preconditioner.cpp:403-409, 936-939), ``Apply``
returns -1 (not implemented in the reference either, :585-592).  Parameters use the
reference's XML key names ("Problem"/"Preconditioner" sublists).
"""
import ctypes as C
import os
import numpy as np

# (HYMLS_MI_LIBRARY: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("HYMLS_MI_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhymls_mi.so")


class HymlsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hymls_mi error %d: %s" % (code, msg))
        self.code = code


class _Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nx", "ny", "nz", "dim", "equations", "dof", "sx", "sy", "sz", "cx", "cy", "cz", "levels",
        "partitioner", "retain_nodes", "retain_pressures", "link_velocities", "link_retained",
        "fix_pressure_level", "nfix")] + [("fix_gid", C.c_int32 * 4), ("variable_type", C.c_int32 * 8),
                                         ("retain_xyz", C.c_int32 * 3), ("retain_at_level", C.c_int32 * 8),
                                         ("retain_at_level_xyz", (C.c_int32 * 3) * 8), ("periodic", C.c_int32 * 3)]


_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_F64P = C.POINTER(C.c_double)
_libs = {}

# transport callbacks of a sharded run (include/hymls_mi.h: hymls_mi_comm)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, _I64P, C.c_void_p, _I64P, C.c_int32, C.c_int32)
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int64)


class _Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("size", C.c_int32), ("alltoallv", A2A_FN), ("alloc", ALLOC_FN)]


def load_library(path=None):
    """Load the C-ABI library.  The product path is hymls_amd/libhymls_mi.so (HIP);
    tests of the host logic pass the path of the test-only simulator explicitly."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). hymls_amd has no CPU fallback." % path)
    if path == os.path.abspath(LIB_PATH) and not os.environ.get("HYMLS_MI_NO_TORCH"):
        # PyTorch-ROCm owns device memory/streams in this process: let it bring up its HIP
        # runtime first so that both sides share one libamdhip64.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    H = C.c_void_p
    sig = {
        "hymls_mi_default_params": (None, [C.POINTER(_Params)]),
        "hymls_mi_create": (C.c_int, [C.POINTER(H), C.POINTER(_Params), C.c_int]),
        "hymls_mi_set_matrix_csr": (C.c_int, [H, C.c_int64, _I32P, _I32P, _F64P]),
        "hymls_mi_set_comm": (C.c_int, [H, C.POINTER(_Comm), C.c_int, C.c_int, C.c_int]),
        "hymls_mi_rank_grid": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "hymls_mi_device_alloc": (C.c_void_p, [H, C.c_int64]),
        "hymls_mi_device_free": (None, [H, C.c_void_p]),
        "hymls_mi_copy_to_host": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int64]),
        "hymls_mi_copy_to_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int64]),
        "hymls_mi_rccl_unique_id": (C.c_int, [C.c_char_p]),
        "hymls_mi_rccl_comm_init": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
        "hymls_mi_rccl_comm_destroy": (None, [C.c_void_p]),
        "hymls_mi_set_comm_rccl": (C.c_int, [H, C.c_void_p, C.c_int, C.c_int, C.c_int]),
        "hymls_mi_comm_selftest": (C.c_int, [H]),
        "hymls_mi_invert_blocks": (C.c_int, [H, C.c_int32, C.c_int32, _F64P]),
        "hymls_mi_required_rows": (C.c_int, [H, _I64P, _I32P]),
        "hymls_mi_set_matrix_rows": (C.c_int, [H, C.c_int64, _I32P, _I32P, _I32P, _F64P]),
        "hymls_mi_owned_rows": (C.c_int, [H, _I64P, _I32P]),
        "hymls_mi_generate_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int64, _I32P,
                                             _I64P, _I32P, _I32P, _F64P]),
        "hymls_mi_set_testvector": (C.c_int, [H, _F64P]),
        "hymls_mi_initialize": (C.c_int, [H]),
        "hymls_mi_compute": (C.c_int, [H]),
        "hymls_mi_apply_inverse": (C.c_int, [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
        "hymls_mi_set_border": (C.c_int, [H, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
        "hymls_mi_apply_inverse_bordered": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        "hymls_mi_apply": (C.c_int, [H, _F64P, _F64P]),
        "hymls_mi_matvec": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int]),
        "hymls_mi_is_initialized": (C.c_int, [H]),
        "hymls_mi_is_computed": (C.c_int, [H]),
        "hymls_mi_num_initialize": (C.c_int, [H]),
        "hymls_mi_num_compute": (C.c_int, [H]),
        "hymls_mi_num_apply_inverse": (C.c_int, [H]),
        "hymls_mi_initialize_time": (C.c_double, [H]),
        "hymls_mi_compute_time": (C.c_double, [H]),
        "hymls_mi_apply_inverse_time": (C.c_double, [H]),
        "hymls_mi_num_levels": (C.c_int, [H]),
        "hymls_mi_level_size": (C.c_int64, [H, C.c_int]),
        "hymls_mi_level_schur_size": (C.c_int64, [H, C.c_int]),
        "hymls_mi_level_num_subdomains": (C.c_int64, [H, C.c_int]),
        "hymls_mi_apply_bytes": (C.c_double, [H, C.c_int]),
        "hymls_mi_setup_flops": (C.c_double, [H, C.c_int]),
        "hymls_mi_last_apply_seconds": (C.c_double, [H, C.c_int]),
        "hymls_mi_set_profiling": (C.c_int, [H, C.c_int]),
        "hymls_mi_stream": (C.c_void_p, [H]),
        "hymls_mi_get_interior": (C.c_int, [H, C.c_int, C.c_int, _I32P, _I32P]),
        "hymls_mi_get_separator_groups": (C.c_int, [H, C.c_int, C.c_int, _I32P, _I32P, _I32P, _I32P, _I32P]),
        "hymls_mi_generate_matrix": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I32P, _I32P, _F64P]),
        "hymls_mi_generate_problem": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int64,
                                                _I32P, _I64P, _I32P, _I32P, _F64P]),
        "hymls_mi_generate_problem_periodic": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                                         C.c_int64, _I32P, _I64P, _I32P, _I32P, _F64P]),
        "hymls_mi_generate_testvector": (C.c_int, [C.c_int64, _I32P, _I32P, _F64P, _F64P]),
        "hymls_mi_drop_by_value": (C.c_int, [C.c_int64, _I32P, _I32P, _F64P, C.c_double, C.c_int, _I64P, _I32P, _I32P, _F64P]),
        "hymls_mi_last_error": (C.c_char_p, [H]),
        "hymls_mi_destroy": (None, [H]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    lib._hymls_symbols = sorted(sig)
    _libs[path] = lib
    return lib


def _i32(a):
    return a.ctypes.data_as(_I32P)


def _f64(a):
    return a.ctypes.data_as(_F64P)


def generate_matrix(equations, nx, ny, nz, a=None, b=1.0, lib=None):
    """CSR arrays (rowptr, colind, val) of the synthetic benchmark matrices:
    'Laplace' (Galeri Laplace3D scaled by -1) or 'Stokes-C' (GaleriExt::Stokes3D with
    a = nx^2, b = 1 by default, reference src/HYMLS_MainUtils.cpp:260-348)."""
    lib = lib or load_library()
    eq = {"Laplace": 0, "Stokes-C": 1}[equations]
    if a is None:
        a = float(nx * nx)
    n, nnz = C.c_int64(), C.c_int64()
    ierr = lib.hymls_mi_generate_matrix(eq, nx, ny, nz, a, b, C.byref(n), C.byref(nnz), None, None, None)
    if ierr:
        raise HymlsError(ierr, "generate_matrix")
    rowptr = np.empty(n.value + 1, np.int32)
    col = np.empty(nnz.value, np.int32)
    val = np.empty(nnz.value, np.float64)
    lib.hymls_mi_generate_matrix(eq, nx, ny, nz, a, b, C.byref(n), C.byref(nnz), _i32(rowptr), _i32(col), _f64(val))
    return rowptr, col, val


def generate_rows(equations, nx, ny, nz, gids, a=None, b=1.0, lib=None):
    """CSR arrays (rowptr, global colind, val) of the rows `gids` of the same matrices (what one rank
    of a sharded run generates for itself)."""
    lib = lib or load_library()
    eq = {"Laplace": 0, "Stokes-C": 1}[equations]
    if a is None:
        a = float(nx * nx)
    gids = np.ascontiguousarray(gids, dtype=np.int32)
    nnz = C.c_int64()
    ierr = lib.hymls_mi_generate_rows(eq, nx, ny, nz, a, b, gids.size, _i32(gids), C.byref(nnz), None, None, None)
    if ierr:
        raise HymlsError(ierr, "generate_rows")
    rowptr = np.empty(gids.size + 1, np.int32)
    col = np.empty(max(nnz.value, 1), np.int32)
    val = np.empty(max(nnz.value, 1), np.float64)
    lib.hymls_mi_generate_rows(eq, nx, ny, nz, a, b, gids.size, _i32(gids), C.byref(nnz), _i32(rowptr), _i32(col), _f64(val))
    return rowptr, col[:nnz.value], val[:nnz.value]


_PROBLEM = {"Laplace": 0, "Stokes-C": 1, "Stokes": 1, "Darcy": 2, "Oseen": 3, "Cavity": 3}


def generate_problem(problem, nx, ny, nz, a=None, b=None, re=0.0, gids=None, lib=None, periodic=(False, False, False)):
    """CSR arrays (rowptr, global colind, val) of a BASELINE input (include/hymls_mi.h: hymls_mi_generate_problem):
    'Laplace', 'Stokes' (a = nx^2, b = 1), 'Darcy' (a = 1, b = -1, reference src/HYMLS_MainUtils.cpp:300-306),
    'Cavity' (Stokes + convection at Reynolds number `re`).  gids: only these rows (sharded runs)."""
    lib = lib or load_library()
    kind = _PROBLEM[problem]
    if a is None:
        a = 1.0 if kind == 2 else float(nx * nx)
    if b is None:
        b = -1.0 if kind == 2 else 1.0
    n = nx * ny * nz * (1 if kind == 0 else 4)
    gp = None
    if gids is not None:
        gids = np.ascontiguousarray(gids, dtype=np.int32)
        n, gp = gids.size, _i32(gids)
    nnz = C.c_int64()
    per = sum((4 >> d) for d in range(3) if periodic[d])       # GaleriExt::PERIO_Flag bits
    ierr = lib.hymls_mi_generate_problem_periodic(kind, nx, ny, nz, a, b, re, per, n, gp, C.byref(nnz), None, None, None)
    if ierr:
        raise HymlsError(ierr, "generate_problem")
    rowptr = np.empty(n + 1, np.int32)
    col = np.empty(max(nnz.value, 1), np.int32)
    val = np.empty(max(nnz.value, 1), np.float64)
    lib.hymls_mi_generate_problem_periodic(kind, nx, ny, nz, a, b, re, per, n, gp, C.byref(nnz), _i32(rowptr), _i32(col), _f64(val))
    return rowptr, col[:nnz.value], val[:nnz.value]


def generate_testvector_rows(gids, rowptr, col, val):
    """create_testvector for a list of rows with global column ids (0 for rows that only have a
    diagonal entry, 1 otherwise; reference src/HYMLS_MainUtils.cpp:208-258)."""
    gids = np.asarray(gids)
    rowptr = np.asarray(rowptr)
    n = gids.size
    tv = np.zeros(n)
    if n == 0 or rowptr[-1] == 0:
        return tv
    off = (val != 0.0) & (col != np.repeat(gids, np.diff(rowptr)))
    nonempty = np.flatnonzero(np.diff(rowptr) > 0)      # (reduceat needs strictly increasing starts)
    tv[nonempty] = np.logical_or.reduceat(off, rowptr[nonempty])
    return tv


def generate_testvector(rowptr, col, val, lib=None):
    lib = lib or load_library()
    tv = np.empty(rowptr.size - 1, np.float64)
    lib.hymls_mi_generate_testvector(rowptr.size - 1, _i32(rowptr), _i32(col), _f64(val), _f64(tv))
    return tv


def drop_by_value(A, tol=1e-14, kind="RelDropDiag", lib=None):
    """MatrixUtils::DropByValue as Compute applies it (include/hymls_mi.h: hymls_mi_drop_by_value); A a scipy matrix"""
    import scipy.sparse as sp
    lib = lib or load_library()
    A = A.tocsr()
    A.sort_indices()
    rp, ci, va = (np.ascontiguousarray(A.indptr, np.int32), np.ascontiguousarray(A.indices, np.int32),
                  np.ascontiguousarray(A.data, np.float64))
    k = {"RelDropDiag": 0, "RelZeroDiag": 1, "RelFullDiag": 2}[kind]
    nnz = C.c_int64()
    ierr = lib.hymls_mi_drop_by_value(A.shape[0], _i32(rp), _i32(ci), _f64(va), tol, k, C.byref(nnz), None, None, None)
    if ierr:
        raise HymlsError(ierr, "drop_by_value")
    orp, oci, ova = np.empty(A.shape[0] + 1, np.int32), np.empty(max(nnz.value, 1), np.int32), np.empty(max(nnz.value, 1))
    lib.hymls_mi_drop_by_value(A.shape[0], _i32(rp), _i32(ci), _f64(va), tol, k, C.byref(nnz), _i32(orp), _i32(oci), _f64(ova))
    return sp.csr_matrix((ova[:nnz.value], oci[:nnz.value], orp), shape=A.shape)


_EQ = {"Laplace": 0, "Stokes-C": 1}
_PART = {"Cartesian": 0, "Skew Cartesian": 1}
_VT = {"Laplace": 0, "Velocity U": 1, "Velocity V": 2, "Velocity W": 3, "Pressure": 4, "Interior": 5}


class Preconditioner:
    """HYMLS::Preconditioner(K, params, testVector) on one MI355X.

    K       : (rowptr, colind, val) CSR arrays or a scipy.sparse matrix, rows in GID order
    params  : dict with the reference's sublists, e.g.
              {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": 32, "ny": 32, "nz": 32},
               "Preconditioner": {"Separator Length": 8, "Number of Levels": 1, "Partitioner": "Cartesian"}}
    """

    def __init__(self, K, params, testVector=None, device=0, lib=None, comm=None, rank_grid=None):
        """comm / rank_grid: sharded run (one process per GPU): `comm` is a hymls_amd.dist.TorchComm,
        rank_grid = (px, py, pz) boxes of the grid; K may then be None (call RequiredRows(), then
        SetMatrixRows(gids, (rowptr, global colind, val)) with this rank's rows)."""
        self._lib = load_library(lib) if (lib is None or isinstance(lib, str)) else lib
        self._h = C.c_void_p()
        self._params = params
        p = _Params()
        self._lib.hymls_mi_default_params(C.byref(p))
        self._fill(p, params)
        ierr = self._lib.hymls_mi_create(C.byref(self._h), C.byref(p), device)
        self._check(ierr)
        self._n = None
        self._comm = comm
        if comm is not None:
            comm.attach(self)
            px, py, pz = rank_grid
            if getattr(comm, "nccl_comm", None) is not None:   # built-in RCCL transport (hymls_amd.dist.RcclComm)
                self._check(self._lib.hymls_mi_set_comm_rccl(self._h, comm.nccl_comm, px, py, pz))
            else:
                self._check(self._lib.hymls_mi_set_comm(self._h, C.byref(comm.c_struct), px, py, pz))
        if K is not None:
            self.SetMatrix(K)
        if testVector is not None:
            self.SetTestVector(testVector)

    # --- sharded runs
    def CommSelfTest(self):
        """collective: 0 if the transport of this sharded handle moved a stamped all-to-all correctly, else the error code"""
        return self._lib.hymls_mi_comm_selftest(self._h)

    def InvertBlocks(self, blocks):
        """in-place inverses of a stack of dense blocks, shape (nblk, nb, nb) (the separator-block step of Compute on its
        own: Ifpack_DenseContainer in the reference, src/HYMLS_SchurPreconditioner.cpp:284-291)"""
        B = np.asarray(blocks, dtype=np.float64)
        assert B.ndim == 3 and B.shape[1] == B.shape[2]
        cm = np.array(B.transpose(0, 2, 1), dtype=np.float64, order="C", copy=True)     # column-major blocks (never the caller's memory)
        self._check(self._lib.hymls_mi_invert_blocks(self._h, B.shape[1], B.shape[0], cm.ctypes.data_as(_F64P)))
        return np.ascontiguousarray(cm.transpose(0, 2, 1))

    def RequiredRows(self):
        n = C.c_int64()
        self._check(self._lib.hymls_mi_required_rows(self._h, C.byref(n), None))
        g = np.empty(n.value, np.int32)
        self._check(self._lib.hymls_mi_required_rows(self._h, C.byref(n), _i32(g)))
        return g

    def SetMatrixRows(self, gids, K):
        gids = np.ascontiguousarray(gids, dtype=np.int32)
        rowptr = np.ascontiguousarray(K[0], dtype=np.int32)
        col = np.ascontiguousarray(K[1], dtype=np.int32)
        val = np.ascontiguousarray(K[2], dtype=np.float64)
        self._check(self._lib.hymls_mi_set_matrix_rows(self._h, gids.size, _i32(gids), _i32(rowptr), _i32(col), _f64(val)))
        return 0

    def SetTestVector(self, tv):
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        self._check(self._lib.hymls_mi_set_testvector(self._h, _f64(tv)))

    def OwnedRows(self):
        n = C.c_int64()
        self._check(self._lib.hymls_mi_owned_rows(self._h, C.byref(n), None))
        g = np.empty(n.value, np.int32)
        self._check(self._lib.hymls_mi_owned_rows(self._h, C.byref(n), _i32(g)))
        self._n = int(n.value)
        return g

    @staticmethod
    def _fill(p, params):
        prob = params.get("Problem", {})
        prec = params.get("Preconditioner", {})
        p.dim = prob.get("Dimension", 3)
        p.nx = prob.get("nx", -1)
        p.ny = prob.get("ny", -1)
        p.nz = prob.get("nz", -1)
        if "Equations" in prob:
            if prob["Equations"] not in _EQ:
                raise HymlsError(-2, "'Equations' parameter not recognized")
            p.equations = _EQ[prob["Equations"]]
        else:
            p.equations = -1
            p.dof = prob.get("Degrees of Freedom", 1)
            for d in range(p.dof):
                vt = prob.get("Variable %d" % d, {}).get("Variable Type", "Laplace")
                if vt == "Velocity":
                    vt = "Velocity " + "UVW"[sum(1 for e in range(d) if prob.get("Variable %d" % e, {}).get(
                        "Variable Type", "Laplace").startswith("Velocity"))]
                p.variable_type[d] = _VT[vt]
        p.sx = prec.get("Separator Length (x)", prec.get("Separator Length", 4))
        p.sy = prec.get("Separator Length (y)", prec.get("Separator Length", -1))
        p.sz = prec.get("Separator Length (z)", prec.get("Separator Length", -1))
        p.cx = prec.get("Coarsening Factor (x)", prec.get("Coarsening Factor", -1))
        p.cy = prec.get("Coarsening Factor (y)", prec.get("Coarsening Factor", -1))
        p.cz = prec.get("Coarsening Factor (z)", prec.get("Coarsening Factor", -1))
        p.levels = prec.get("Number of Levels", 1)
        p.partitioner = _PART[prec.get("Partitioner", "Cartesian")]
        p.retain_nodes = prec.get("Retain Nodes", -1)
        for d, ax in enumerate("xyz"):
            p.retain_xyz[d] = prec.get("Retain Nodes (%s)" % ax, -1)
        for l in range(8):
            p.retain_at_level[l] = prec.get("Retain Nodes at Level %d" % l, -1)
            for d, ax in enumerate("xyz"):
                p.retain_at_level_xyz[l][d] = prec.get("Retain Nodes at Level %d (%s)" % (l, ax), -1)
        # GaleriExt::PERIO_Flag bits: X 4, Y 2, Z 1 (reference src/GaleriExt_Periodic.h)
        perio = int(prob.get("Periodicity", sum((4 >> d) for d, ax in enumerate("xyz") if prob.get("%s-periodic" % ax, False))))
        for d in range(3):
            p.periodic[d] = (perio >> (2 - d)) & 1
        p.retain_pressures = prob.get("Retained Pressure Nodes", -1)
        p.link_velocities = int(prec.get("Eliminate Velocities Together", True))
        p.link_retained = int(prec.get("Eliminate Retained Nodes Together", True))
        p.fix_pressure_level = int(prec.get("Fix Pressure Level", True))
        fix = []
        k = 1
        while ("Fix GID %d" % k) in prec:
            fix.append(prec["Fix GID %d" % k])
            k += 1
        p.nfix = len(fix)
        for i, g in enumerate(fix[:4]):
            p.fix_gid[i] = g
        variant = prec.get("Preconditioner Variant", "Block Diagonal")
        if variant != "Block Diagonal":
            raise HymlsError(-99, "Variant '%s' not implemented" % variant)

    def _check(self, ierr):
        if ierr != 0:
            raise HymlsError(ierr, self._lib.hymls_mi_last_error(self._h).decode())

    # --- reference API
    def SetMatrix(self, K):
        if hasattr(K, "tocsr"):
            K = K.tocsr()
            K.sort_indices()
            K = (K.indptr, K.indices, K.data)
        rowptr = np.ascontiguousarray(K[0], dtype=np.int32)
        col = np.ascontiguousarray(K[1], dtype=np.int32)
        val = np.ascontiguousarray(K[2], dtype=np.float64)
        self._n = rowptr.size - 1
        self._check(self._lib.hymls_mi_set_matrix_csr(self._h, self._n, _i32(rowptr), _i32(col), _f64(val)))
        return 0

    def Initialize(self):
        self._check(self._lib.hymls_mi_initialize(self._h))
        if self._comm is not None:
            self.OwnedRows()
        return 0

    def Compute(self):
        self._check(self._lib.hymls_mi_compute(self._h))
        if self._comm is not None:
            self.OwnedRows()
        return 0

    def IsInitialized(self):
        return bool(self._lib.hymls_mi_is_initialized(self._h))

    def IsComputed(self):
        return bool(self._lib.hymls_mi_is_computed(self._h))

    def ApplyInverse(self, B, X=None):
        """X = P^{-1} B.  numpy arrays (n,) or (n, nvec) column-major semantics; or torch
        CUDA tensors (float64, contiguous columns) for the device-resident path."""
        if hasattr(B, "data_ptr"):  # torch tensor on the device
            import torch
            if X is None:
                X = torch.empty_like(B)
            nvec = 1 if B.dim() == 1 else B.shape[0]
            # device multivectors are stored as (nvec, n) row-major = column-major (n, nvec)
            assert B.dtype == torch.float64 and B.is_contiguous() and X.is_contiguous()
            self._check(self._lib.hymls_mi_apply_inverse(self._h, B.data_ptr(), self._n, X.data_ptr(), self._n, nvec, 1))
            return X
        Bc = np.asfortranarray(np.asarray(B, dtype=np.float64))
        Xc = np.empty_like(Bc, order="F")
        nvec = 1 if Bc.ndim == 1 else Bc.shape[1]
        self._check(self._lib.hymls_mi_apply_inverse(self._h, Bc.ctypes.data, self._n, Xc.ctypes.data, self._n, nvec, 0))
        if X is not None:
            X[...] = Xc
            return X
        return Xc

    # --- BorderedOperator (reference src/HYMLS_BorderedOperator.hpp)
    def SetBorder(self, V, W=None, C_=None):
        """[K V; W' C]: V, W (n, m) arrays, C (m, m); W defaults to V, C to zero; V=None removes the border.
        Compute() has to be called afterwards (reference src/HYMLS_Preconditioner.cpp:844-918)."""
        if V is None:
            self._check(self._lib.hymls_mi_set_border(self._h, 0, None, 0, None, 0, None))
            self._m = 0
            return 0
        V = np.asfortranarray(np.asarray(V, dtype=np.float64).reshape(self._n, -1))
        m = V.shape[1]
        W = None if W is None else np.asfortranarray(np.asarray(W, dtype=np.float64).reshape(self._n, m))
        Cm = None if C_ is None else np.asfortranarray(np.asarray(C_, dtype=np.float64).reshape(m, m))
        self._check(self._lib.hymls_mi_set_border(self._h, m, V.ctypes.data, self._n, W.ctypes.data if W is not None else None,
                                                  self._n, Cm.ctypes.data if Cm is not None else None))
        self._m = m
        return 0

    def HaveBorder(self):
        return getattr(self, "_m", 0) > 0

    def ApplyInverseBordered(self, B, T):
        """ApplyInverse(B, T, X, S): returns (X, S) with [K V; W' C] [X; S] ~ [B; T]; B a host array or a device tensor."""
        m = getattr(self, "_m", 0)
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(m))
        S = np.zeros(m)
        if hasattr(B, "data_ptr"):
            import torch
            X = torch.empty_like(B)
            self._check(self._lib.hymls_mi_apply_inverse_bordered(self._h, B.data_ptr(), T.ctypes.data, X.data_ptr(), S.ctypes.data, 1))
            return X, S
        Bc = np.ascontiguousarray(B, dtype=np.float64)
        X = np.empty_like(Bc)
        self._check(self._lib.hymls_mi_apply_inverse_bordered(self._h, Bc.ctypes.data, T.ctypes.data, X.ctypes.data, S.ctypes.data, 0))
        return X, S

    def Apply(self, X, Y):
        return -1  # not implemented in the reference either

    def MatVec(self, X, Y=None):
        if hasattr(X, "data_ptr"):
            import torch
            if Y is None:
                Y = torch.empty_like(X)
            self._check(self._lib.hymls_mi_matvec(self._h, X.data_ptr(), Y.data_ptr(), 1))
            return Y
        Xc = np.ascontiguousarray(X, dtype=np.float64)
        Yc = np.empty_like(Xc)
        self._check(self._lib.hymls_mi_matvec(self._h, Xc.ctypes.data, Yc.ctypes.data, 0))
        return Yc

    def SetUseTranspose(self, flag):
        return -1

    def UseTranspose(self):
        return False

    def HasNormInf(self):
        return False

    def Condest(self):
        return -1.0

    def Label(self):
        return "Preconditioner"

    def NumInitialize(self):
        return self._lib.hymls_mi_num_initialize(self._h)

    def NumCompute(self):
        return self._lib.hymls_mi_num_compute(self._h)

    def NumApplyInverse(self):
        return self._lib.hymls_mi_num_apply_inverse(self._h)

    def InitializeTime(self):
        return self._lib.hymls_mi_initialize_time(self._h)

    def ComputeTime(self):
        return self._lib.hymls_mi_compute_time(self._h)

    def ApplyInverseTime(self):
        return self._lib.hymls_mi_apply_inverse_time(self._h)

    # --- measurement / introspection
    def level_sizes(self):
        n = self._lib.hymls_mi_num_levels(self._h)
        return [(l, self._lib.hymls_mi_level_size(self._h, l), self._lib.hymls_mi_level_schur_size(self._h, l),
                 self._lib.hymls_mi_level_num_subdomains(self._h, l)) for l in range(n)]

    def apply_bytes(self, which=0):
        return self._lib.hymls_mi_apply_bytes(self._h, which)

    def setup_flops(self, which=0):
        return self._lib.hymls_mi_setup_flops(self._h, which)

    def set_profiling(self, on=True):
        self._lib.hymls_mi_set_profiling(self._h, int(on))

    def last_apply_seconds(self, which=0):
        return self._lib.hymls_mi_last_apply_seconds(self._h, which)

    def stream(self):
        return self._lib.hymls_mi_stream(self._h)

    def interior(self, level, sd):
        n = C.c_int32()
        self._check(self._lib.hymls_mi_get_interior(self._h, level, sd, C.byref(n), None))
        out = np.empty(n.value, np.int32)
        self._lib.hymls_mi_get_interior(self._h, level, sd, C.byref(n), _i32(out))
        return out

    def separator_groups(self, level, sd):
        """[(type, owned, nodes)] of every separator group around subdomain sd."""
        ng = C.c_int32()
        self._check(self._lib.hymls_mi_get_separator_groups(self._h, level, sd, C.byref(ng), None, None, None, None))
        gptr = np.zeros(ng.value + 1, np.int32)
        self._lib.hymls_mi_get_separator_groups(self._h, level, sd, C.byref(ng), _i32(gptr), None, None, None)
        gtype = np.empty(ng.value, np.int32)
        owned = np.empty(ng.value, np.int32)
        nodes = np.empty(max(int(gptr[-1]), 1), np.int32)
        self._lib.hymls_mi_get_separator_groups(self._h, level, sd, C.byref(ng), _i32(gptr), _i32(gtype), _i32(owned), _i32(nodes))
        return [(int(gtype[g]), bool(owned[g]), nodes[gptr[g]:gptr[g + 1]].copy()) for g in range(ng.value)]

    def close(self):
        if self._h:
            self._lib.hymls_mi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
