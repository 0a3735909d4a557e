"""HYMLS preconditioner setup + apply (oracle side).  TEST INFRASTRUCTURE ONLY.

A serial numpy/scipy restatement of the reference algorithm *as the reference
performs it* (explicit dense Schur parts per subdomain, two-sided Householder
per separator group, dropping by pattern, dense block LU, recursion).  The
product (hymls_amd) restructures the arithmetic for the GPU; this file is what
its results are compared against.

Reference:
  src/HYMLS_Preconditioner.cpp:279-517    Initialize / Compute
  src/HYMLS_Preconditioner.cpp:930-1070   ApplyInverse
  src/HYMLS_Preconditioner.cpp:781-818    CreateTestVector
  src/HYMLS_MatrixBlock.cpp:74-385        blocks, subdomain solvers
  src/HYMLS_SchurComplement.cpp:88-306    Construct / Construct11 / Construct22
  src/HYMLS_SchurPreconditioner.cpp:182-340,384-629,698-986,1010-1093,1236-1265,1311-1346,1435-1459
  src/HYMLS_Householder.cpp:38-163,353-369
  src/HYMLS_RestrictedOT.hpp:21-37
  src/HYMLS_CoarseSolver.cpp:101-323
  src/HYMLS_MatrixUtils.cpp:1011-1309     DropByValue / PutDirichlet
  src/HYMLS_Macros.hpp:26-30              HYMLS_SMALL_ENTRY = 1e-14
LU factorisations: scipy (SuperLU / LAPACK getrf) -- values unpinned by the
reference's tests (any backward-stable LU).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla

from .partition import Params, HierarchicalMap

SMALL = 1e-14  # HYMLS_SMALL_ENTRY


def _sign(x):
    # src/HYMLS_Householder.cpp:15-18  (sign(0) = 0)
    return -1.0 if x < 0 else (1.0 if x > 0 else 0.0)


def householder_apply_rows(X, v):
    """Householder::Apply(X, v): X = (2uu'/u'u - I) X in place on the rows of X
    (src/HYMLS_Householder.cpp:38-80)."""
    v = v * _sign(v[0])
    nrmv = np.linalg.norm(v)
    v1 = v[0] + nrmv
    if abs(v1) < SMALL or nrmv < SMALL:
        return
    fac1 = 1.0 / (nrmv * v1)
    fac2 = nrmv * X[0, :] + v @ X
    fac = fac1 * fac2
    u = v.copy()
    u[0] = v1
    X[:, :] = np.outer(u, fac) - X


def restricted_ot(Sk, pos, v):
    """RestrictedOT::Apply (src/HYMLS_RestrictedOT.hpp:21-37): rows then columns."""
    n = v.size
    if n <= 0:
        return
    householder_apply_rows(Sk[pos:pos + n, :], v)
    householder_apply_rows(Sk[:, pos:pos + n].T, v)


def householder_row(v):
    """Householder::Construct (src/HYMLS_Householder.cpp:128-163): the normalised
    vector stored as one sparse row of T; None if its norm is < SMALL."""
    v = v * _sign(v[0])
    nrm = np.linalg.norm(v)
    v = v.copy()
    v[0] = v[0] + nrm
    nrm = np.linalg.norm(v)
    if nrm < SMALL:
        return None
    return v / nrm


# Switch for comparisons with the product under ITS rule (off = the reference's literal DropByValue): a diagonal entry
# at rounding level, |a_ii| <= droptol x the largest entry of row i, counts as the structural zero it is on paper (the
# relative threshold of the reference's ComputeScaling, src/HYMLS_SparseDirectSolver.cpp:632-664).  The product applies
# it in its drop_by_value (hymls_amd/csrc/precond.cpp) because the pressure diagonals of a reduced matrix, which cancel
# exactly on the CPU, come out as ~1e-14 x rowmax from the GPU's summation order and must not be taken for velocities by
# the zero-diagonal test of the orderings (src/HYMLS_MatrixUtils.cpp:1344-1352).  The rule only fits droptol =
# HYMLS_SMALL_ENTRY: any larger tolerance would drop legitimately small diagonals.
ROUNDING_LEVEL_DIAGONAL_IS_ZERO = False


def drop_by_value(A, droptol=SMALL, kind="RelZeroDiag", rounding_level_diagonal_is_zero=None):
    """MatrixUtils::DropByValue (src/HYMLS_MatrixUtils.cpp:1011-1212)."""
    A = A.tocsr()
    n = A.shape[0]
    if ROUNDING_LEVEL_DIAGONAL_IS_ZERO if rounding_level_diagonal_is_zero is None else rounding_level_diagonal_is_zero:
        A = A.copy()
        A.sort_indices()
        rows_ = np.repeat(np.arange(n), np.diff(A.indptr))
        rmax = np.zeros(n)
        np.maximum.at(rmax, rows_, np.abs(A.data))
        tiny = (rows_ == A.indices) & (np.abs(A.data) <= droptol * rmax[rows_])
        A.data[tiny] = 0.0
    rel = kind in ("Relative", "RelDropDiag", "RelZeroDiag", "RelFullDiag")
    abs_diag = kind in ("RelDropDiag", "RelZeroDiag", "RelFullDiag", "AbsZeroDiag", "AbsFullDiag", "Absolute")
    zero_diag = kind in ("RelZeroDiag", "AbsZeroDiag")
    full_diag = kind in ("RelFullDiag", "AbsFullDiag")
    diag = A.diagonal()
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    cols = A.indices
    vals = A.data
    is_diag = rows == cols
    scal = np.ones(vals.size)
    if rel:
        scal = np.maximum(np.abs(diag[rows]), np.abs(diag[cols]))
    if abs_diag:
        scal = np.where(is_diag, 1.0, scal)
    keep = (np.abs(vals) > scal * droptol) & (np.abs(vals) > droptol)
    r, c, v = rows[keep], cols[keep], vals[keep]
    if zero_diag:
        m = is_diag & ~keep
        r = np.concatenate([r, rows[m]])
        c = np.concatenate([c, cols[m]])
        v = np.concatenate([v, np.zeros(m.sum())])
    if full_diag:
        m = r != c
        dv = np.where(np.abs(diag) > droptol, diag, 0.0)
        r = np.concatenate([r[m], np.arange(n)])
        c = np.concatenate([c[m], np.arange(n)])
        v = np.concatenate([v[m], dv])
    # explicit zeros (zero_diag / full_diag) are kept: scipy does not prune them here
    res = sp.csr_matrix((v, (r, c)), shape=A.shape)
    res.sort_indices()
    return res


class CoarseSolver:
    """src/HYMLS_CoarseSolver.cpp:131-323 (plain, un-bordered path)."""

    def __init__(self, S, gids, fix_gids, permc_spec=None):
        self.gids = np.asarray(gids)
        S = drop_by_value(S, SMALL, "RelFullDiag").tolil()
        self.fix_lids = []
        for g in fix_gids:
            w = np.nonzero(self.gids == g)[0]
            if w.size == 0:
                raise RuntimeError("fix GID %d not in matrix row map" % g)
            lid = int(w[0])
            # PutDirichlet (MatrixUtils.cpp:1229-1309): zero row & column, unit diagonal
            S[lid, :] = 0.0
            S[:, lid] = 0.0
            S[lid, lid] = 1.0
            self.fix_lids.append(lid)
        self.S = S.tocsc()
        self.n = self.S.shape[0]
        # (permc_spec: SuperLU column ordering; the values of the exact LU are unpinned by the reference's tests)
        self.lu = (spla.splu(self.S, permc_spec=permc_spec) if permc_spec else spla.splu(self.S)) if self.n else None
        self.border = None

    def set_border(self, V, W, C):
        """CoarseSolver::SetBorder + the bordered branch of Compute (src/HYMLS_CoarseSolver.cpp:196-260,425-445):
        the border goes explicitly into an AugmentedMatrix [S V; W' C] that is factored by the direct solver."""
        m = V.shape[1]
        aug = sp.bmat([[self.S, sp.csr_matrix(V)], [sp.csr_matrix(W.T), sp.csr_matrix(C)]]).tocsc()
        self.border = (m, spla.splu(aug))

    def apply_inverse_bordered(self, x, T):
        """CoarseSolver::ApplyInverse(X, T, Y, S) (:454-560): full augmented right-hand side, no Dirichlet zeroing."""
        m, lu = self.border
        sol = lu.solve(np.concatenate([x, T]))
        return sol[:self.n], sol[self.n:]

    def apply_inverse(self, x):
        if self.n == 0:
            return x.copy()
        rhs = np.array(x, dtype=float, copy=True)
        for lid in self.fix_lids:
            if lid > 0:  # CoarseSolver.cpp:288-289 (lid 0 is not zeroed)
                rhs[lid] = 0.0
        return self.lu.solve(rhs)


class SchurPreconditioner:
    """src/HYMLS_SchurPreconditioner.cpp (Block Diagonal variant, dropping + OT on)."""

    def __init__(self, prec, testvec2):
        self.prec = prec
        hm = prec.hm
        self.n2 = prec.map2.size
        pos2 = prec.pos2
        self.tv2 = testvec2
        # InitializeOT (:384-467): one row of T per owned group
        rows, cols, vals = [], [], []
        self.vsum_pos = []
        r = 0
        for sd in range(hm.nsd):
            for gi in hm.owned[sd]:
                idx = pos2[hm.groups[sd][gi][1]]
                self.vsum_pos.append(idx[0])
                w = householder_row(testvec2[idx])
                if w is not None:
                    rows.append(np.full(idx.size, r))
                    cols.append(idx)
                    vals.append(w)
                r += 1
        self.vsum_pos = np.array(self.vsum_pos, dtype=np.int64)
        if rows:
            self.T = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                   shape=(r, self.n2))
        else:
            self.T = sp.csr_matrix((r, self.n2))

    def apply_ot(self, x):
        """ApplyOT (:1236-1265) = Householder::Apply(Tv,T,v): 2 T'(T v) - v."""
        return 2.0 * (self.T.T @ (self.T @ x)) - x

    def compute(self):
        prec = self.prec
        hm = prec.hm
        pos2 = prec.pos2
        n2 = self.n2
        r22, c22, v22 = [], [], []
        r11, c11, v11 = [], [], []
        for sd in range(hm.nsd):
            seps = hm.sd_separators(sd)
            if seps.size == 0:
                continue
            gpos = pos2[seps]
            v = self.tv2[gpos]
            for part, (R, C, V) in (("22", (r22, c22, v22)), ("11", (r11, c11, v11))):
                Sk = prec.construct22(sd) if part == "22" else prec.construct11(sd)
                # ConstructSCPart (:877-986)
                pos = 0
                starts = []
                for (_, g) in hm.groups[sd]:
                    restricted_ot(Sk, pos, v[pos:pos + g.size])
                    starts.append(pos)
                    pos += g.size
                starts = np.array(starts)
                vs = gpos[starts]
                R.append(np.repeat(vs, vs.size))
                C.append(np.tile(vs, vs.size))
                V.append(Sk[np.ix_(starts, starts)].ravel())
                for L in hm.linked[sd]:
                    loc = np.concatenate([np.arange(starts[gi] + 1, starts[gi] + hm.groups[sd][gi][1].size)
                                          for gi in L]) if L else np.zeros(0, int)
                    if loc.size == 0:
                        continue
                    gl = gpos[loc]
                    R.append(np.repeat(gl, gl.size))
                    C.append(np.tile(gl, gl.size))
                    V.append(Sk[np.ix_(loc, loc)].ravel())
        def cat(x, dt):
            return np.concatenate(x) if x else np.zeros(0, dt)
        r22, c22, v22 = cat(r22, np.int64), cat(c22, np.int64), cat(v22, float)
        r11, c11, v11 = cat(r11, np.int64), cat(c11, np.int64), cat(v11, float)
        # A22 part: Replace (identical values from every subdomain) -> keep first occurrence
        key = r22 * n2 + c22
        _, first = np.unique(key, return_index=True)
        M22 = sp.csr_matrix((v22[first], (r22[first], c22[first])), shape=(n2, n2))
        M11 = sp.csr_matrix((v11, (r11, c11)), shape=(n2, n2))  # SumInto: duplicates summed
        self.matrix = (M22 + M11).tocsr()
        # block solvers (InitializeBlocks :301-340 + Compute :284-291)
        self.blocks = []
        for sd in range(hm.nsd):
            for L in hm.owned_linked(sd):
                ids = np.concatenate([pos2[hm.groups[sd][gi][1][1:]] for gi in L])
                if ids.size == 0:
                    continue
                D = self.matrix[ids][:, ids].toarray()
                self.blocks.append((ids, sla.lu_factor(D)))
        # ComputeNextLevel (:520-629)
        vs = self.vsum_pos
        reduced = self.matrix[vs][:, vs]
        reduced = drop_by_value(reduced, SMALL, "RelDropDiag")
        self.reduced = reduced
        vs_gids = prec.map2[vs]
        border = getattr(self, "border", None)
        if border is not None:
            # ComputeBorder (:631-664): transform V and W, their V-sum rows are the border of the next level
            SV, SW, SC = border
            self.bV = np.column_stack([self.apply_ot(SV[:, j]) for j in range(SV.shape[1])])
            self.bW = np.column_stack([self.apply_ot(SW[:, j]) for j in range(SW.shape[1])])
        if prec.level + 1 < prec.max_level:
            next_tv = self.apply_ot(self.tv2)[vs]
            self.next = Preconditioner(reduced, prec.params.next_level(), level=prec.level + 1,
                                       gids=vs_gids, testvector=next_tv, ngid=prec.ngid)
            if border is not None:
                self.next.set_border(self.bV[vs], self.bW[vs], SC)
            self.next.compute()
        else:
            self.next = CoarseSolver(reduced, vs_gids, prec.params.fix_gids)
            if border is not None:
                self.next.set_border(self.bV[vs], self.bW[vs], SC)

    def set_border(self, V, W, C):
        self.border = (V, W, C)

    def apply_inverse_bordered(self, x, T):
        """bordered ApplyInverse (:1517-1617): block-diagonal solve, T - W'(M11 \ f1), bordered next level."""
        B = self.apply_ot(x)
        Y = np.zeros_like(B)
        for ids, lu in self.blocks:
            Y[ids] = sla.lu_solve(lu, B[ids])
        vs = self.vsum_pos
        Tc = T - self.bW.T @ Y
        Y[vs], S = self.next.apply_inverse_bordered(B[vs], Tc)
        return self.apply_ot(Y), S

    def apply_inverse(self, x):
        """ApplyInverse (:1010-1093)."""
        B = self.apply_ot(x)
        Y = np.zeros_like(B)
        for ids, lu in self.blocks:
            Y[ids] = sla.lu_solve(lu, B[ids])
        vs = self.vsum_pos
        Y[vs] = self.next.apply_inverse(B[vs])
        return self.apply_ot(Y)


class Preconditioner:
    """HYMLS::Preconditioner, serial, un-bordered (src/HYMLS_Preconditioner.cpp).

    A        : scipy sparse (n x n); row i belongs to GID gids[i]
    params   : partition.Params (finalized) for THIS level
    level    : myLevel_ ; params.levels = XML "Number of Levels" (maxLevel_)
    """

    def __init__(self, A, params: Params, level=0, gids=None, testvector=None, ngid=None):
        self.A = A.tocsr()
        self.params = params
        self.level = level
        self.max_level = params.levels
        n = A.shape[0]
        self.gids = np.arange(n, dtype=np.int64) if gids is None else np.asarray(gids, dtype=np.int64)
        self.ngid = params.nx * params.ny * params.nz * params.dof if ngid is None else ngid
        self.testvector = np.ones(n) if testvector is None else np.asarray(testvector, dtype=float)
        self.computed = False
        self._initialize()

    def _initialize(self):
        g2l = np.full(self.ngid, -1, dtype=np.int64)
        g2l[self.gids] = np.arange(self.gids.size)
        present = g2l >= 0
        self.g2l = g2l
        self.hm = HierarchicalMap(self.params, present=None if present.all() else present)
        hm = self.hm
        self.map1 = hm.interior_map()
        self.map2 = hm.separator_map()
        self.pos2 = np.full(self.ngid, -1, dtype=np.int64)
        self.pos2[self.map2] = np.arange(self.map2.size)
        self.i1 = g2l[self.map1]
        self.i2 = g2l[self.map2]
        both = np.concatenate([self.map1, self.map2])
        if both.size != self.gids.size or not np.array_equal(np.sort(both), np.sort(self.gids)):
            raise RuntimeError("partition does not cover the map exactly once")

    def compute(self):
        A, hm = self.A, self.hm
        self.A12 = A[self.i1][:, self.i2].tocsr()
        self.A21 = A[self.i2][:, self.i1].tocsr()
        self.A22 = A[self.i2][:, self.i2].tocsr()
        # subdomain solvers (MatrixBlock::ComputeSubdomainSolvers)
        self.lu = []
        self.loc1 = []
        off = 0
        for sd in range(hm.nsd):
            li = self.g2l[hm.interior[sd]]
            self.loc1.append(np.arange(off, off + li.size))
            off += li.size
            if li.size:
                self.lu.append(spla.splu(A[li][:, li].tocsc()))
            else:
                self.lu.append(None)
        tv2 = self.testvector[self.i2]  # CreateTestVector (:781-818)
        sb = None
        if getattr(self, "border", None) is not None:
            # ComputeBorder (:519-588): border of the Schur complement system
            V, W, C = self.border
            V1, V2, W1, W2 = V[self.i1], V[self.i2], W[self.i1], W[self.i2]
            self.W1 = W1
            self.Q1 = np.column_stack([self.a11_inverse(V1[:, j]) for j in range(V.shape[1])])
            SV = V2 - self.A21 @ self.Q1
            SW = W2 - self.A12.T @ np.column_stack([self.a11_inverse(W1[:, j], trans=True) for j in range(W.shape[1])])
            sb = (SV, SW, C - W1.T @ self.Q1)
        if self.level >= self.max_level:
            # Preconditioner.cpp:485-500: explicit SC, direct solve
            S = self.A22.tolil(copy=True).tocsr()
            rr, cc, vv = [], [], []
            for sd in range(hm.nsd):
                seps = hm.sd_separators(sd)
                if seps.size == 0 or hm.interior[sd].size == 0:
                    continue
                Sk = self.construct11(sd)
                gp = self.pos2[seps]
                rr.append(np.repeat(gp, gp.size))
                cc.append(np.tile(gp, gp.size))
                vv.append(Sk.ravel())
            if rr:
                S = S + sp.csr_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))),
                                      shape=S.shape)
            S = drop_by_value(S, SMALL, "RelZeroDiag")
            self.schur = CoarseSolver(S, self.map2, self.params.fix_gids)
            if sb is not None:
                self.schur.set_border(*sb)
        else:
            self.schur = SchurPreconditioner(self, tv2)
            if sb is not None:
                self.schur.set_border(*sb)
            self.schur.compute()
        self.computed = True
        return self

    # --- SchurComplement::Construct11 / Construct22 (src/HYMLS_SchurComplement.cpp:131-306)
    def construct22(self, sd):
        seps = self.hm.sd_separators(sd)
        li = self.g2l[seps]
        return self.A[li][:, li].toarray()

    def construct11(self, sd):
        seps = self.hm.sd_separators(sd)
        ls = self.g2l[seps]
        li = self.g2l[self.hm.interior[sd]]
        if li.size == 0:
            return np.zeros((ls.size, ls.size))
        A12 = self.A[li][:, ls].toarray()
        A21 = self.A[ls][:, li]
        B = self.lu[sd].solve(A12)
        return -(A21 @ B)

    def a11_inverse(self, b1, trans=False):
        x1 = np.empty_like(b1)
        for sd, lu in enumerate(self.lu):
            if lu is not None:
                x1[self.loc1[sd]] = lu.solve(b1[self.loc1[sd]], trans="T" if trans else "N")
        return x1

    def set_border(self, V, W=None, C=None):
        """BorderedOperator::SetBorder (src/HYMLS_Preconditioner.cpp:844-918): [K V; W' C]; W defaults to V, C to 0.
        Compute has to be called afterwards."""
        V = np.asarray(V, dtype=float).reshape(self.A.shape[0], -1)
        W = V if W is None else np.asarray(W, dtype=float).reshape(self.A.shape[0], -1)
        m = V.shape[1]
        C = np.zeros((m, m)) if C is None else np.asarray(C, dtype=float).reshape(m, m)
        self.border = (V, W, C)
        self.computed = False

    def apply_inverse_bordered(self, b, T):
        """ApplyInverse(B, T, X, S) (src/HYMLS_Preconditioner.cpp:930-1070 with the border branches)."""
        if not self.computed:
            raise RuntimeError("The preconditioner has not yet been computed.")
        b = np.asarray(b, dtype=float)
        x1 = self.a11_inverse(b[self.i1])
        rhs2 = b[self.i2] - self.A21 @ x1
        q = np.asarray(T, dtype=float) - self.W1.T @ x1
        x2, S = self.schur.apply_inverse_bordered(rhs2, q)
        x1 = x1 - self.a11_inverse(self.A12 @ x2) - self.Q1 @ S
        x = np.zeros_like(b)
        x[self.i1] = x1
        x[self.i2] = x2
        return x, S

    def apply_inverse(self, b):
        """Preconditioner::ApplyInverse (src/HYMLS_Preconditioner.cpp:930-1070)."""
        if not self.computed:
            raise RuntimeError("The preconditioner has not yet been computed.")
        b = np.asarray(b, dtype=float)
        b1 = b[self.i1]
        b2 = b[self.i2]
        x1 = self.a11_inverse(b1)
        y2 = self.A21 @ x1
        x2 = self.schur.apply_inverse(b2 - y2)
        y1 = self.A12 @ x2
        x1 = x1 - self.a11_inverse(y1)
        x = np.zeros_like(b)
        x[self.i1] = x1
        x[self.i2] = x2
        return x

    def level_sizes(self):
        out = [(self.level, self.A.shape[0], self.map2.size)]
        s = self.schur
        if isinstance(s, SchurPreconditioner):
            if isinstance(s.next, Preconditioner):
                out += s.next.level_sizes()
            else:
                out.append((self.level + 1, s.next.n, 0))
        return out
