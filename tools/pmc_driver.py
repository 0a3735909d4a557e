"""PMC driver without torch (rocprofv3 --pmc crashes with the torch runtime in the process):
ctypes only, host vectors.  usage: python3 tools/pmc_driver.py N SX LEVELS NAPPLY"""
import os, sys
os.environ["HYMLS_MI_NO_TORCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hymls_amd
n, sx, levels, napply = (int(a) for a in sys.argv[1:5])
rp, ci, va = hymls_amd.generate_matrix("Stokes-C", n, n, n)
tv = hymls_amd.generate_testvector(rp, ci, va)
prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
       "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv)
P.Compute()
b = np.random.default_rng(0).uniform(-1, 1, rp.size - 1)
for _ in range(napply):
    x = P.ApplyInverse(b)
lv = P.level_sizes()
print("PMCDRIVER n %d levels %s bytes_interior_per_launch %.0f n1 %d" % (n, lv, P.apply_bytes(1) / 2 + 16.0 * (lv[0][1] - lv[0][2]), lv[0][1] - lv[0][2]))
