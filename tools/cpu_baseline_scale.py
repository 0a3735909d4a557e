"""CPU baseline on the benched problem family AT SCALE: the compiled CPU oracle
(oracle/cpu/hymls_cpu.cpp: the reference's algorithm as the reference performs it, OpenMP over subdomains = the
reference's unit of parallelism, src/HYMLS_MatrixBlock.cpp:311-385) on GaleriExt Stokes3D N^3, 3-level, Skew Cartesian,
separator length 8 -- the configuration of bench.py's default line, at N = 128 by default (8.4 M DoF; the 256^3 problem
needs ~8 x the memory and setup time).  ApplyInverse is timed at 1 thread (= one MPI rank of the reference) and at the
host cores a GPU box grants per GPU; right-preconditioned GMRES to 1e-8 gives the iteration count.
  python3 tools/cpu_baseline_scale.py [N=128] [threads=16] [out.json]
TEST / MEASUREMENT INFRASTRUCTURE: runs the oracle, never the product."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import galeri, cpu_oracle, krylov
from oracle.partition import Params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = sys.argv[3] if len(sys.argv) > 3 else None
levels = int(os.environ.get("CPU_BASELINE_LEVELS", "2"))
t0 = time.time()
lib = cpu_oracle.load(cpu_oracle.build(native=True, out_dir=tempfile.mkdtemp(prefix="hymls_cpu_")))
A = galeri.stokes3d(n, n, n)
tv = galeri.create_testvector(A)
print("matrix %d^3: %d DoF, %d nonzeros, %.1f s" % (n, A.shape[0], A.nnz, time.time() - t0), flush=True)
p = Params(nx=n, ny=n, nz=n, sx=8, levels=levels, equations="Stokes-C", partitioner="Skew Cartesian").finalize()
t0 = time.time()
O = cpu_oracle.Preconditioner(A, p, testvector=tv, nthreads=threads, lib=lib).compute()
t_setup = time.time() - t0
print("setup (partition in Python + Compute on %d threads): %.1f s, levels %s, nnz(L+U) %d" % (threads, t_setup, O.level_sizes(), O.nnz_factors()), flush=True)
rng = np.random.default_rng(0)
b = rng.uniform(-1, 1, A.shape[0])
res = {"workload": "GaleriExt Stokes3D %d^3 (%d DoF), HYMLS %d-level, Skew Cartesian sx=8" % (n, A.shape[0], levels + 1),
       "levels": O.level_sizes(), "nnz_factors": int(O.nnz_factors()), "setup_s": t_setup, "kind": "port",
       "host_cpus_visible": len(os.sched_getaffinity(0)), "rates": {}}
for nt in (1, threads):
    O.set_threads(nt)
    O.apply_inverse(b)
    reps, t0 = 0, time.time()
    while reps < 2 or (time.time() - t0 < 20.0 and reps < 30):
        O.apply_inverse(b)
        reps += 1
    dt = (time.time() - t0) / reps
    res["rates"][str(nt)] = {"threads": nt, "seconds_per_apply": dt, "dof_per_s": A.shape[0] / dt, "applies": reps}
    print("ApplyInverse at %d thread(s): %.3f s = %.2f MDoF/s (%d applies)" % (nt, dt, A.shape[0] / dt / 1e6, reps), flush=True)
O.set_threads(threads)
x_ex = rng.uniform(-1, 1, A.shape[0])
rhs = A @ x_ex
t0 = time.time()
_, its, rr = krylov.gmres(lambda v: A @ v, rhs, O.apply_inverse, tol=1e-8, maxit=400, restart=100)
res["krylov"] = {"method": "GMRES(100), right preconditioned, zero initial guess, b = K x_ex", "iterations": int(its),
                 "true_relative_residual": float(rr), "seconds": time.time() - t0}
print("GMRES(100): %d iterations, residual %.2e, %.1f s" % (its, rr, time.time() - t0), flush=True)
print(json.dumps(res))
if out:
    json.dump(res, open(out, "w"), indent=1)
