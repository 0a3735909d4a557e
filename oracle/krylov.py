"""Minimal Krylov callers for iteration-count checks.  TEST INFRASTRUCTURE ONLY.

Stands in for the Belos loop the reference drives (src/HYMLS_BaseSolver.cpp:119-139,
309-397): restarted GMRES with left or right preconditioning and preconditioned CG,
convergence on ||r||/||b|| <= tol.
"""
import numpy as np


def gmres(matvec, b, prec=None, tol=1e-8, maxit=200, restart=None, right=True, x0=None):
    n = b.size
    prec = prec or (lambda v: v)
    restart = restart or maxit
    x = np.zeros(n) if x0 is None else x0.copy()
    bnorm = np.linalg.norm(b if right else prec(b))
    if bnorm == 0:
        return x, 0, 0.0
    its = 0
    while its < maxit:
        r = b - matvec(x)
        if not right:
            r = prec(r)
        beta = np.linalg.norm(r)
        if beta / bnorm <= tol:
            break
        m = min(restart, maxit - its)
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 1, m))
        cs = np.zeros(m)
        sn = np.zeros(m)
        g = np.zeros(m + 1)
        g[0] = beta
        V[0] = r / beta
        k = 0
        for k in range(m):
            w = matvec(prec(V[k])) if right else prec(matvec(V[k]))
            for i in range(k + 1):
                H[i, k] = w @ V[i]
                w = w - H[i, k] * V[i]
            H[k + 1, k] = np.linalg.norm(w)
            if H[k + 1, k] > 0:
                V[k + 1] = w / H[k + 1, k]
            for i in range(k):
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            d = np.hypot(H[k, k], H[k + 1, k])
            cs[k], sn[k] = H[k, k] / d, H[k + 1, k] / d
            H[k, k] = d
            H[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            its += 1
            if abs(g[k + 1]) / bnorm <= tol:
                break
        y = np.linalg.solve(np.triu(H[:k + 1, :k + 1]), g[:k + 1])
        dx = V[:k + 1].T @ y
        x = x + (prec(dx) if right else dx)
        if abs(g[k + 1]) / bnorm <= tol:
            break
    res = np.linalg.norm(b - matvec(x)) / np.linalg.norm(b)
    return x, its, res


def pcg(matvec, b, prec=None, tol=1e-8, maxit=200, x0=None):
    prec = prec or (lambda v: v)
    x = np.zeros(b.size) if x0 is None else x0.copy()
    r = b - matvec(x)
    z = prec(r)
    p = z.copy()
    rz = r @ z
    bnorm = np.linalg.norm(b)
    its = 0
    while its < maxit and np.linalg.norm(r) / bnorm > tol:
        Ap = matvec(p)
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        z = prec(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
        its += 1
    return x, its, np.linalg.norm(b - matvec(x)) / bnorm
