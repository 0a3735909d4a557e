"""Compiled CPU oracle: the multilevel preconditioner with the numerics of every level in oracle/cpu/hymls_cpu.cpp
(OpenMP over subdomains, sparse subdomain factors -- see its header).  TEST INFRASTRUCTURE ONLY: the checker of
parity tests at sizes oracle/hymls.py cannot reach in test time, and the CPU baseline bench.py times beside the GPU path.

Same structure as oracle/hymls.py (which stays the primary restatement and pins this one: tests/test_cpu_oracle.py):
partition and recursion in Python (oracle/partition.py, oracle/skew.py: reference src/HYMLS_OverlappingPartitioner.cpp,
src/HYMLS_HierarchicalMap.cpp, src/HYMLS_SchurPreconditioner.cpp:520-629 ComputeNextLevel), DropByValue and the coarse
solver (src/HYMLS_CoarseSolver.cpp:131-323, an exact sparse LU: SuperLU here, serial as the reference's Amesos KLU)
from oracle/hymls.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

from .hymls import SMALL, CoarseSolver, drop_by_value
from .partition import HierarchicalMap, Params

# column ordering of the coarse SuperLU: scipy's default (COLAMD).  Measured on the 48^3 / 64^3 two-level Stokes samples:
# MMD_AT_PLUS_A with SuperLU's partial pivoting fills the saddle-point coarse matrix 10 x more (setup 94 s instead of 17 s).
COARSE_ORDERING = None

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu")
_LIB = None
_I = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)


def build(native=False, out_dir=None):
    """g++ -O3 -fopenmp (native: -march=native into out_dir, for timing on the machine it runs on)."""
    if not native:
        subprocess.check_call(["make", "-s", "-C", _DIR])
        return os.path.join(_DIR, "libhymls_cpu_oracle.so")
    out = os.path.join(out_dir or "/tmp", "libhymls_cpu_oracle_native.so")
    subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-o", out,
                           os.path.join(_DIR, "hymls_cpu.cpp")])
    return out


def load(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or os.path.join(_DIR, "libhymls_cpu_oracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.hcpu_level_create.restype = C.c_void_p
    lib.hcpu_level_create.argtypes = [C.c_int, _I, _I, _D, C.c_int, _I, _I, _I, _I, _I, _I, _I, _I, _D, C.c_int, C.c_int]
    lib.hcpu_level_destroy.argtypes = [C.c_void_p]
    lib.hcpu_level_flags.argtypes = [C.c_void_p]
    lib.hcpu_level_sizes.argtypes = [C.c_void_p, _I, _I, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.hcpu_level_reduced.argtypes = [C.c_void_p, _I, _I, _D, _I, _D]
    lib.hcpu_level_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.hcpu_level_apply_pre.argtypes = [C.c_void_p, _D, _D]
    lib.hcpu_level_apply_post.argtypes = [C.c_void_p, _D, _D]
    lib.hcpu_max_threads.restype = C.c_int
    _LIB = lib
    return lib


def _i(a):
    return a.ctypes.data_as(_I)


def _d(a):
    return a.ctypes.data_as(_D)


class Preconditioner:
    """Mirror of oracle.hymls.Preconditioner (un-bordered): compute(), apply_inverse(b), level_sizes()."""

    def __init__(self, A, params: Params, level=0, gids=None, testvector=None, ngid=None, nthreads=1, lib=None):
        self.lib = lib or load()
        self.A = sp.csr_matrix(A)
        self.A.sort_indices()
        self.params = params
        self.level = level
        self.max_level = params.levels
        self.nthreads = nthreads
        n = self.A.shape[0]
        self.n = n
        self.gids = np.arange(n, dtype=np.int64) if gids is None else np.asarray(gids, dtype=np.int64)
        self.ngid = params.nx * params.ny * params.nz * params.dof if ngid is None else ngid
        self.testvector = np.ones(n) if testvector is None else np.asarray(testvector, dtype=float)
        self.h = None
        self.next = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hcpu_level_destroy(self.h)
            self.h = None

    def compute(self):
        g2l = np.full(self.ngid, -1, dtype=np.int64)
        g2l[self.gids] = np.arange(self.gids.size)
        present = g2l >= 0
        hm = HierarchicalMap(self.params, present=None if present.all() else present)
        self.hm = hm
        int_ptr = np.zeros(hm.nsd + 1, np.int32)
        gsd_ptr = np.zeros(hm.nsd + 1, np.int32)
        int_idx, g_ptr, g_idx, g_owned, g_link, g_olink = [], [0], [], [], [], []
        for sd in range(hm.nsd):
            int_idx.append(g2l[hm.interior[sd]])
            int_ptr[sd + 1] = int_ptr[sd] + hm.interior[sd].size
            ng = len(hm.groups[sd])
            gsd_ptr[sd + 1] = gsd_ptr[sd] + ng
            link = np.full(ng, -1, np.int32)
            for li, L in enumerate(hm.linked[sd]):
                link[L] = li
            olink = np.full(ng, -1, np.int32)
            for li, L in enumerate(hm.owned_linked(sd)):
                olink[L] = li
            owned = np.zeros(ng, np.int32)
            owned[hm.owned[sd]] = 1
            for gi, (_, g) in enumerate(hm.groups[sd]):
                g_idx.append(g2l[g])
                g_ptr.append(g_ptr[-1] + g.size)
            g_owned.append(owned); g_link.append(link); g_olink.append(olink)
        cat = lambda x, dt: np.ascontiguousarray(np.concatenate(x) if x else np.zeros(0), dtype=dt)
        int_idx, g_idx = cat(int_idx, np.int32), cat(g_idx, np.int32)
        g_owned, g_link, g_olink = cat(g_owned, np.int32), cat(g_link, np.int32), cat(g_olink, np.int32)
        g_ptr = np.asarray(g_ptr, np.int32)
        A = self.A
        rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data, dtype=np.float64)
        tv = np.ascontiguousarray(self.testvector, dtype=np.float64)
        direct = self.level >= self.max_level
        self.direct = direct
        if self.h:
            self.lib.hcpu_level_destroy(self.h)
        self.h = self.lib.hcpu_level_create(self.n, _i(rp), _i(ci), _d(va), hm.nsd, _i(int_ptr), _i(int_idx), _i(gsd_ptr), _i(g_ptr),
                                            _i(g_idx), _i(g_owned), _i(g_link), _i(g_olink), _d(tv), int(direct), self.nthreads)
        if not self.h:
            raise RuntimeError("partition does not cover the map exactly once")
        self.flags = self.lib.hcpu_level_flags(self.h)
        n1, n2, nred = C.c_int32(), C.c_int32(), C.c_int32()
        rnnz, lunnz = C.c_int64(), C.c_int64()
        self.lib.hcpu_level_sizes(self.h, C.byref(n1), C.byref(n2), C.byref(nred), C.byref(rnnz), C.byref(lunnz))
        self.n1, self.n2, self.nred, self.nnz_lu = n1.value, n2.value, nred.value, lunnz.value
        ptr = np.empty(nred.value + 1, np.int32); col = np.empty(rnnz.value, np.int32); val = np.empty(rnnz.value)
        nodes = np.empty(nred.value, np.int32); ntv = np.empty(nred.value)
        self.lib.hcpu_level_reduced(self.h, _i(ptr), _i(col), _d(val), _i(nodes), _d(ntv))
        R = sp.csr_matrix((val, col, ptr), shape=(nred.value, nred.value))
        red_gids = self.gids[nodes]
        if direct:
            # Preconditioner.cpp:485-500: S assembled, DropByValue (RelZeroDiag), CoarseSolver
            self.next = CoarseSolver(drop_by_value(R, SMALL, "RelZeroDiag"), red_gids, self.params.fix_gids, permc_spec=COARSE_ORDERING)
        else:
            R = drop_by_value(R, SMALL, "RelDropDiag")       # SchurPreconditioner.cpp:548-549
            if self.level + 1 < self.max_level:
                self.next = Preconditioner(R, self.params.next_level(), level=self.level + 1, gids=red_gids,
                                           testvector=ntv, ngid=self.ngid, nthreads=self.nthreads, lib=self.lib).compute()
            else:
                self.next = CoarseSolver(R, red_gids, self.params.fix_gids, permc_spec=COARSE_ORDERING)
        self.vr = np.empty(max(self.nred, 1))
        self.x = np.empty(self.n)
        return self

    def set_threads(self, nthreads):
        self.nthreads = nthreads
        self.lib.hcpu_level_set_threads(self.h, nthreads)
        if isinstance(self.next, Preconditioner):
            self.next.set_threads(nthreads)

    def apply_inverse(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        self.lib.hcpu_level_apply_pre(self.h, _d(b), _d(self.vr))
        vs = np.ascontiguousarray(self.next.apply_inverse(self.vr[:self.nred]), dtype=np.float64) if self.nred else self.vr
        x = np.empty(self.n)
        self.lib.hcpu_level_apply_post(self.h, _d(vs), _d(x))
        return x

    def level_sizes(self):
        out = [(self.level, self.n, self.n2)]
        if isinstance(self.next, Preconditioner):
            out += self.next.level_sizes()
        elif not self.direct:
            out.append((self.level + 1, self.next.n, 0))
        return out

    def nnz_factors(self):
        t = self.nnz_lu
        if isinstance(self.next, Preconditioner):
            t += self.next.nnz_factors()
        return t
