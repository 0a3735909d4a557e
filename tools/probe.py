"""ad-hoc perf probe (not part of the product): time Initialize/Compute/ApplyInverse."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hymls_amd
eq, n, sx, levels = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
part = (sys.argv[5] if len(sys.argv) > 5 else "Cartesian").replace("_", " ")
t = time.time(); rp, ci, va = hymls_amd.generate_matrix(eq, n, n, n); tv = hymls_amd.generate_testvector(rp, ci, va); tg = time.time() - t
prm = {"Problem": {"Equations": eq, "Dimension": 3, "nx": n, "ny": n, "nz": n},
       "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": part}}
P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv)
t = time.time(); P.Initialize(); ti = time.time() - t
t = time.time(); P.Compute(); tc = time.time() - t
N = rp.size - 1
b = torch.rand(N, dtype=torch.float64, device="cuda") * 2 - 1
x = torch.empty_like(b)
for _ in range(3): P.ApplyInverse(b, x)
torch.cuda.synchronize()
K = 20
t = time.time()
for _ in range(K): P.ApplyInverse(b, x)
torch.cuda.synchronize(); ta = (time.time() - t) / K
P.set_profiling(True); P.ApplyInverse(b, x); torch.cuda.synchronize()
ph = [P.last_apply_seconds(i) for i in range(4)]
P.set_profiling(False)
r = b - P.MatVec(x)
print(json.dumps({"eq": eq, "n": n, "sx": sx, "levels": levels, "N": N, "gen_s": tg, "init_s": ti, "compute_s": tc,
  "apply_ms": ta * 1e3, "DoF_per_s": N / ta, "bytes": [P.apply_bytes(i) for i in range(6)], "GBps": P.apply_bytes(0) / ta / 1e9,
  "phases_ms": [p * 1e3 for p in ph], "levels_info": P.level_sizes(), "resid_after_1_apply": float(r.norm() / b.norm())}))
