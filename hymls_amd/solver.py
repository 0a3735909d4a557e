"""Device-resident Krylov caller of the preconditioner: the semantics of the reference's
``HYMLS::BaseSolver`` (src/HYMLS_BaseSolver.hpp:49-215, src/HYMLS_BaseSolver.cpp:119-139 SetPrecond,
:300-397 ApplyInverse) with the Belos managers it drives replaced by two small loops on torch tensors:
restarted GMRES ("Krylov Method" = "GMRES", Belos BlockGmres with block size 1) and preconditioned CG.
Every vector stays in HBM; per iteration the host sees two scalars.

Parameter names are the reference's ("Solver" sublist of the XML files, e.g. testSuite/integration_tests/*.xml):
    {"Krylov Method": "GMRES" | "CG", "Initial Vector": "Zero" | "Random" | "Previous",
     "Left or Right Preconditioning": "Right" | "Left",
     "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 500,
                          "Num Blocks": 250, "Maximum Restarts": 20}}
Sharded runs (the operator and the preconditioner are sharded hymls_amd.Preconditioner objects): every rank
calls the solver with its part of B; inner products are all-reduced over torch.distributed, K x imports the
columns owned by other ranks (hymls_mi_matvec), ApplyInverse exchanges as described in DESIGN.md section 6.
"""
import numpy as np
import torch
import torch.distributed as dist


class Solver:
    def __init__(self, K, precond=None, params=None):
        """K: object with MatVec(x) (a hymls_amd.Preconditioner applies the matrix it was given) or a callable;
        precond: object with ApplyInverse(b) or None."""
        self.SetOperator(K)
        self.SetPrecond(precond)
        self._numIter = 0
        self._achieved = float("nan")
        self._prev = None
        self.setParameterList(params or {})

    # --- reference API
    def setParameterList(self, params):
        sol = params.get("Solver", params)
        it = sol.get("Iterative Solver", {})
        self.method = sol.get("Krylov Method", "GMRES")
        if self.method not in ("GMRES", "CG"):
            raise ValueError("Krylov Method must be GMRES or CG")
        self.start = sol.get("Initial Vector", "Zero")
        self.lor = sol.get("Left or Right Preconditioning", "Right")
        self.tol = float(it.get("Convergence Tolerance", 1e-8))
        self.maxit = int(it.get("Maximum Iterations", 500))
        self.restart = int(it.get("Num Blocks", 250))
        self.max_restarts = int(it.get("Maximum Restarts", 20))
        self.seed = sol.get("Random Seed", 1234)

    def SetOperator(self, K):
        self._matvec = K.MatVec if hasattr(K, "MatVec") else K
        self._sharded = getattr(K, "_comm", None) is not None

    # --- reductions (all-reduced over the ranks of a sharded run)
    def _reduce(self, t):
        if self._sharded:
            dist.all_reduce(t)
        return t

    def _dot(self, a, b):
        return float(self._reduce(torch.dot(a, b).reshape(1)))

    def _norm(self, a):
        return float(np.sqrt(self._dot(a, a)))

    def _nonzero(self, x):
        return self._dot(x, x) > 0.0

    def SetPrecond(self, P):
        self._prec = (P.ApplyInverse if hasattr(P, "ApplyInverse") else P)

    def SetTolerance(self, tol):
        self.tol = float(tol)

    def getNumIter(self):
        return self._numIter

    def achievedTol(self):
        return self._achieved

    def ApplyMatrix(self, x):
        return self._matvec(x).clone()

    def ApplyPrec(self, x):
        return self._prec(x).clone() if self._prec is not None else x.clone()

    def ApplyInverse(self, B, X=None):
        """solve K X = B; B: device tensor (n,) float64.  Returns X; getNumIter() gives the iteration count.
        Raises RuntimeError if the tolerance was not reached (the reference returns a nonzero code and warns)."""
        if self.start == "Random":
            g = torch.Generator(device=B.device); g.manual_seed(self.seed)
            x0 = torch.rand(B.numel(), dtype=B.dtype, device=B.device, generator=g) * 2 - 1
        elif self.start == "Previous" and self._prev is not None:
            x0 = self._prev.clone()
        else:
            x0 = torch.zeros_like(B)
        x, its, rel = (self._gmres if self.method == "GMRES" else self._cg)(B, x0)
        self._numIter, self._achieved, self._prev = its, rel, x
        if X is not None:
            X.copy_(x)
            x = X
        if not rel <= self.tol:
            raise RuntimeError("Krylov solver did not converge: %d iterations, relative residual %.3e" % (its, rel))
        return x

    # --- GMRES(m), right or left preconditioned, classical Gram-Schmidt applied twice, Givens rotations
    def _gmres(self, b, x0):
        n, m = b.numel(), min(self.restart, self.maxit)
        right = self.lor == "Right"
        V = torch.empty((m + 1, n), dtype=b.dtype, device=b.device)
        x = x0
        its, rel, beta0 = 0, float("inf"), None
        for _cycle in range(self.max_restarts + 1):
            r = b - self.ApplyMatrix(x) if (its > 0 or self._nonzero(x)) else b.clone()
            if not right:
                r = self.ApplyPrec(r)
            beta = self._norm(r)
            if beta0 is None:
                beta0 = beta
            if beta0 == 0.0:
                return x, its, 0.0
            rel = beta / beta0
            if rel <= self.tol or its >= self.maxit:
                break
            V[0] = r / beta
            H = np.zeros((m + 1, m))
            cs, sn = np.zeros(m), np.zeros(m)
            gvec = np.zeros(m + 1); gvec[0] = beta
            k_used = 0
            for k in range(m):
                w = self.ApplyMatrix(self.ApplyPrec(V[k])) if right else self.ApplyPrec(self.ApplyMatrix(V[k]))
                Vk = V[:k + 1]
                h = self._reduce(torch.mv(Vk, w)); w -= torch.mv(Vk.t(), h)
                h2 = self._reduce(torch.mv(Vk, w)); w -= torch.mv(Vk.t(), h2)
                H[:k + 1, k] = (h + h2).cpu().numpy()
                H[k + 1, k] = self._norm(w)
                if H[k + 1, k] > 0:
                    V[k + 1] = w / H[k + 1, k]
                for i in range(k):
                    t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                    H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
                    H[i, k] = t
                d = np.hypot(H[k, k], H[k + 1, k])
                cs[k], sn[k] = H[k, k] / d, H[k + 1, k] / d
                H[k, k] = d; H[k + 1, k] = 0.0
                gvec[k + 1] = -sn[k] * gvec[k]; gvec[k] = cs[k] * gvec[k]
                its += 1; k_used = k + 1
                rel = abs(gvec[k + 1]) / beta0
                if rel <= self.tol or its >= self.maxit:
                    break
            y = np.linalg.solve(np.triu(H[:k_used, :k_used]), gvec[:k_used])
            z = torch.mv(V[:k_used].t(), torch.from_numpy(y).to(b.device))
            x = x + (self.ApplyPrec(z) if right else z)
            if rel <= self.tol or its >= self.maxit:
                break
        return x, its, rel

    # --- preconditioned CG (Belos PseudoBlockCG semantics: relative to the initial residual)
    def _cg(self, b, x0):
        x = x0.clone()
        r = b - self.ApplyMatrix(x) if self._nonzero(x) else b.clone()
        z = self.ApplyPrec(r)
        p = z.clone()
        rz = self._dot(r, z)
        r0 = self._norm(r)
        if r0 == 0.0:
            return x, 0, 0.0
        its, rel = 0, 1.0
        while its < self.maxit:
            Ap = self.ApplyMatrix(p)
            alpha = rz / self._dot(p, Ap)
            x += alpha * p
            r -= alpha * Ap
            its += 1
            rel = self._norm(r) / r0
            if rel <= self.tol:
                break
            z = self.ApplyPrec(r)
            rz_new = self._dot(r, z)
            p = z + (rz_new / rz) * p
            rz = rz_new
        return x, its, rel


class BorderedSolver(Solver):
    """Krylov solve of the bordered system [K V; W' C] [x; s] = [b; t] with the bordered preconditioner
    (the reference's HYMLS::BorderedSolver, src/HYMLS_BorderedSolver.hpp/.cpp: the Krylov vectors are the
    augmented vectors [x; s], the operator applies the border explicitly, the preconditioner solves its own
    bordered system level by level).  One GPU."""

    def __init__(self, K, precond, params=None):
        super().__init__(K, precond, params)
        self._K, self._P = K, precond
        self._V = self._W = self._C = None

    def SetBorder(self, V, W=None, C=None, device=None):
        """also sets the border of the preconditioner; call precond.Compute() afterwards (as in the reference)"""
        if V is None:
            self._V = self._W = self._C = None
            self._P.SetBorder(None)
            return 0
        V = np.asarray(V, dtype=np.float64).reshape(V.shape[0], -1)
        W = V if W is None else np.asarray(W, dtype=np.float64).reshape(V.shape)
        m = V.shape[1]
        C = np.zeros((m, m)) if C is None else np.asarray(C, dtype=np.float64).reshape(m, m)
        self._P.SetBorder(V, W, C)
        self._V = torch.from_numpy(np.ascontiguousarray(V)).to(device) if device else torch.from_numpy(np.ascontiguousarray(V))
        self._W = torch.from_numpy(np.ascontiguousarray(W)).to(self._V.device)
        self._C = torch.from_numpy(np.ascontiguousarray(C)).to(self._V.device)
        return 0

    def ApplyInverse(self, B, T=None, X=None):
        """returns (X, S); B: device tensor (n,), T: array-like (m,) (default 0)"""
        if self._V is None:
            return super().ApplyInverse(B, X), np.zeros(0)
        n, m = B.numel(), self._V.shape[1]
        if self._V.device != B.device:
            self._V, self._W, self._C = self._V.to(B.device), self._W.to(B.device), self._C.to(B.device)
        t = torch.zeros(m, dtype=B.dtype, device=B.device) if T is None else torch.as_tensor(np.asarray(T, dtype=np.float64)).to(B.device)
        kmv, papply = self._K.MatVec if hasattr(self._K, "MatVec") else self._K, self._P.ApplyInverseBordered

        def matvec(z):
            y = torch.empty_like(z)
            y[:n] = kmv(z[:n].contiguous()) + self._V @ z[n:]
            y[n:] = self._W.t() @ z[:n] + self._C @ z[n:]
            return y

        def prec(z):
            x, s = papply(z[:n].contiguous(), z[n:].cpu().numpy())
            return torch.cat([x, torch.from_numpy(s).to(z.device)])

        saved = (self._matvec, self._prec)
        self._matvec, self._prec = matvec, prec
        try:
            z = super().ApplyInverse(torch.cat([B, t]))
        finally:
            self._matvec, self._prec = saved
        if X is not None:
            X.copy_(z[:n])
        return z[:n].clone(), z[n:].cpu().numpy()
