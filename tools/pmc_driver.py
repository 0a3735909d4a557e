"""PMC driver: one process, no torch (hymls_amd imports torch only for the Krylov caller), host vectors.
usage: python3 tools/pmc_driver.py N SX LEVELS NAPPLY [PROBLEM]
Meant to run under `rocprofv3 --pmc <counter> --kernel-include-regex <kernel> --kernel-trace -- python3 tools/pmc_driver.py ...`.
Writes /proc/self/maps next to its output before the first kernel so that a crash inside the tool can be symbolised."""
import faulthandler, os, sys
faulthandler.enable()
os.environ["HYMLS_MI_NO_TORCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hymls_amd
assert "torch" not in sys.modules, "the PMC driver must stay torch-free"
n, sx, levels, napply = (int(a) for a in sys.argv[1:5])
problem = sys.argv[5] if len(sys.argv) > 5 else "Stokes"
out = os.environ.get("PMC_DRIVER_OUT", "gpurun_out")
os.makedirs(out, exist_ok=True)
rp, ci, va = hymls_amd.generate_problem(problem, n, n, n, re=1000.0)
tv = hymls_amd.generate_testvector(rp, ci, va)
prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
       "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv)
open(os.path.join(out, "pmc_driver_maps.txt"), "w").write(open("/proc/self/maps").read())
print("PMCDRIVER created", flush=True)
P.Initialize()
print("PMCDRIVER initialized", flush=True)
import time
for c in range(int(os.environ.get("PMC_DRIVER_COMPUTES", "1"))):
    t0 = time.time()
    P.Compute()
    print("PMCDRIVER computed %d in %.3f s" % (c, time.time() - t0), flush=True)
b = np.random.default_rng(0).uniform(-1, 1, rp.size - 1)
for _ in range(napply):
    x = P.ApplyInverse(b)
lv = P.level_sizes()
n1 = lv[0][1] - lv[0][2]
print("PMCDRIVER n %d levels %s bytes_interior_per_launch %.0f stored %.0f sparse_equivalent %.0f n1 %d" % (
    n, lv, min(P.apply_bytes(1), P.apply_bytes(6)) / 2 + 16.0 * n1, P.apply_bytes(1) / 2, P.apply_bytes(6) / 2, n1), flush=True)
