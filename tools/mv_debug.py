"""isolate a wrong multi-vector kernel: per launcher family force group size 1 and compare the columns with single-vector applies"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
if len(sys.argv) > 1:
    import numpy as np
    import hymls_amd
    from common import problem, xml_params, product_prec, rel_diff
    lib = hymls_amd.load_library()
    for eq, n, sx, lv, cx, part in (("Laplace", 8, 4, 0, -1, "Cartesian"), ("Laplace", 16, 4, 2, 2, "Cartesian"), ("Stokes-C", 16, 8, 1, -1, "Skew Cartesian")):
        A, tv = problem(eq, n)
        P = product_prec(A, tv, xml_params(eq, n, sx, lv, cx=cx, partitioner=part), lib)
        B = np.random.default_rng(1).uniform(-1, 1, (A.shape[0], 4))
        X = P.ApplyInverse(B)
        errs = [rel_diff(X[:, j], P.ApplyInverse(B[:, j].copy())) for j in range(4)]
        print("   ", eq, n, sx, lv, ["%.1e" % e for e in errs], flush=True)
else:
    for env in ({}, {"HYMLS_MI_MV_GROUP_FUSED": "1"}, {"HYMLS_MI_MV_GROUP_LVL": "1"}, {"HYMLS_MI_MV_GROUP_BLK": "1"},
                {"HYMLS_MI_MV_GROUP_FUSED": "2"}, {"HYMLS_MI_MV_GROUP_LVL": "2"}, {"HYMLS_MI_NO_FUSED_SOLVE": "1"},
                {"HYMLS_MI_NO_FUSED_SOLVE": "1", "HYMLS_MI_MV_GROUP_LVL": "1"}):
        print("ENV", env, flush=True)
        subprocess.call([sys.executable, __file__, "run"], env=dict(os.environ, **env))
