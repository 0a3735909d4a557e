"""A/B of two builds of the library in ONE process on one GPU (development aid): both preconditioners are set up side by
side (ctypes loads each .so with RTLD_LOCAL), then ApplyInverse alternates between them, so that clocks and
temperature are the same for both.   python tools/ab_apply.py libA.so libB.so [N=256] [levels=2] [problem=Stokes]
Environment switches that a library reads when a handle is created can differ: AB_ENV_A="K=V,K2=V2" AB_ENV_B=...
(two copies of one .so file give two independent instances)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hymls_amd

libs = [os.path.abspath(a) for a in sys.argv[1:3]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
levels = int(sys.argv[4]) if len(sys.argv) > 4 else 2
problem = sys.argv[5] if len(sys.argv) > 5 else "Stokes"
prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
       "Preconditioner": {"Separator Length": 8, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
P = []
K = tv = None
for q, path in enumerate(libs):
    extra = dict(kv.split("=", 1) for kv in os.environ.get("AB_ENV_" + "AB"[q], "").split(",") if "=" in kv)
    for k in list(os.environ):
        if k.startswith("HYMLS_MI_AB_"):
            del os.environ[k]
    saved = {k: os.environ.get(k) for k in extra}
    os.environ.update(extra)
    lib = hymls_amd.load_library(path)
    if K is None:
        K = hymls_amd.generate_problem(problem, n, n, n, re=1000.0, lib=lib)
        tv = hymls_amd.generate_testvector(*K, lib=lib)
    p = hymls_amd.Preconditioner(K, prm, testVector=tv, lib=lib)
    p.Compute()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p.Compute()                       # numeric Compute with the pattern reused
    torch.cuda.synchronize(); t_re = time.perf_counter() - t0
    print("recompute %.3f s" % t_re, flush=True)
    p.set_profiling(True)
    P.append(p)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    print("ready:", os.path.basename(path), extra, p.level_sizes(), flush=True)
N = K[0].size - 1
g = torch.Generator(device="cuda"); g.manual_seed(1)
b = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
x = [torch.empty_like(b), torch.empty_like(b)]
for _ in range(3):
    for q in (0, 1):
        P[q].ApplyInverse(b, x[q])
torch.cuda.synchronize()
tot = [0.0, 0.0]
rounds, reps = 12, 8
for r in range(rounds):
    for q in ((0, 1) if r % 2 == 0 else (1, 0)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            P[q].ApplyInverse(b, x[q])
        torch.cuda.synchronize(); tot[q] += time.perf_counter() - t0
for q in (0, 1):
    print("%-28s %.3f ms / ApplyInverse   interior solves %.3f ms per launch   coarse %.3f ms" % (
        os.path.basename(libs[q]), 1e3 * tot[q] / (rounds * reps), 1e3 * P[q].last_apply_seconds(1) / 2, 1e3 * P[q].last_apply_seconds(4)), flush=True)
print("max |x_A - x_B| / max |x_A| = %.3e" % float((x[0] - x[1]).abs().max() / x[0].abs().max()))
