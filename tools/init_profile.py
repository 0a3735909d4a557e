"""development aid: time Initialize() alone (the host part of the setup), on the simulator or on a GPU box
    HYMLS_MI_VERBOSE=1 python tools/init_profile.py 128 [levels] [gpu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hymls_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"     # the product library on a GPU box instead of the simulator
lib = hymls_amd.load_library() if gpu else hymls_amd.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "hostsim", "libhymls_mi_hostsim.so"))
rp, ci, va = hymls_amd.generate_problem("Stokes", n, n, n, lib=lib)
tv = hymls_amd.generate_testvector(rp, ci, va, lib=lib)
prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
       "Preconditioner": {"Separator Length": 8, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv, lib=lib)
t0 = time.time(); P.Initialize(); print("Initialize %.2f s" % (time.time() - t0))
