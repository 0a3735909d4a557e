"""per-phase timing of k_interior_fused (HYMLS_MI_FUSED_PROF=1): python tools/fused_phase_profile.py  (needs a GPU)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hymls_amd
n = 128
K = hymls_amd.generate_matrix("Stokes-C", n, n, n); tv = hymls_amd.generate_testvector(*K)
prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n}, "Preconditioner": {"Separator Length": 8, "Number of Levels": 1, "Partitioner": "Skew Cartesian"}}
P = hymls_amd.Preconditioner(K, prm, testVector=tv); P.Compute()
b = torch.rand(K[0].size - 1, dtype=torch.float64, device="cuda")
for _ in range(3): P.ApplyInverse(b)
torch.cuda.synchronize()
os.environ["HYMLS_MI_FUSED_PROF"] = "1"
P.ApplyInverse(b); torch.cuda.synchronize()
