"""development aid (GPU box): configurations bench.py does not expose -- non-cubic grids, 2D, Retain Nodes, a border, no pressure pin.
Each one: Initialize, Compute, right-preconditioned GMRES to 1e-8 (true residual checked)."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hymls_amd

def run(name, problem, nx, ny, nz, prec, dim=3, re=0.0, border=False, max_its=600):
    t0 = time.time()
    try:
        rp, ci, va = hymls_amd.generate_problem(problem, nx, ny, nz, re=re)
        tv = hymls_amd.generate_testvector(rp, ci, va)
        prm = {"Problem": {"Equations": "Stokes-C" if problem != "Laplace" else "Laplace", "Dimension": dim, "nx": nx, "ny": ny, "nz": nz},
               "Preconditioner": dict({"Partitioner": "Skew Cartesian" if problem != "Laplace" else "Cartesian"}, **prec)}
        P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv)
        N = rp.size - 1
        if border:
            # the constant pressure mode as a border (what the reference's BorderedSolver deflates)
            V = np.zeros((N, 1)); V[3::4, 0] = 1.0
            P.SetBorder(V)
        P.Compute()
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        x_ex = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        rhs = P.MatVec(x_ex).clone()
        S = hymls_amd.Solver(P, P, {"Krylov Method": "GMRES", "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": max_its, "Num Blocks": 200, "Maximum Restarts": 5}})
        xs = S.ApplyInverse(rhs)
        res = float((rhs - P.MatVec(xs)).norm() / rhs.norm())
        ok = res < 1e-7 and S.getNumIter() < max_its
        print("%-34s %s  levels %s  its %d  true residual %.2e  %.1f s" % (name, "ok    " if ok else "NOT OK", [l[1] for l in P.level_sizes()], S.getNumIter(), res, time.time() - t0), flush=True)
    except Exception as e:
        print("%-34s FAILED %s" % (name, str(e)[:300]), flush=True)

run("stokes 128x64x32 sx8 L2", "Stokes", 128, 64, 32, {"Separator Length": 8, "Number of Levels": 2})
run("stokes 64x64x128 sx8 L1", "Stokes", 64, 64, 128, {"Separator Length": 8, "Number of Levels": 1})
run("stokes 64^3 retain 2 L2", "Stokes", 64, 64, 64, {"Separator Length": 8, "Number of Levels": 2, "Retain Nodes": 2})
run("stokes 64^3 retain (x)=2 L1", "Stokes", 64, 64, 64, {"Separator Length": 8, "Number of Levels": 1, "Retain Nodes (x)": 2})
run("stokes 64^3 no pressure pin", "Stokes", 64, 64, 64, {"Separator Length": 8, "Number of Levels": 1, "Fix Pressure Level": False})
run("stokes 64^3 border L2", "Stokes", 64, 64, 64, {"Separator Length": 8, "Number of Levels": 2}, border=True)
run("darcy 96x96x48 sx8 L2", "Darcy", 96, 96, 48, {"Separator Length": 8, "Number of Levels": 2})
run("cavity 64^3 re 2000 sx16 L1", "Cavity", 64, 64, 64, {"Separator Length": 16, "Number of Levels": 1}, re=2000.0)
run("laplace 128^3 sx8 L2", "Laplace", 128, 128, 128, {"Separator Length": 8, "Number of Levels": 2})
run("laplace 128^3 sx4 L3 cx4", "Laplace", 128, 128, 128, {"Separator Length": 4, "Number of Levels": 3, "Coarsening Factor": 4})
