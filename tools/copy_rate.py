"""Host -> device and device -> host copy rate of the library's own copy path (hymls_mi_copy_to_device / _to_host) on a
2 GiB pageable buffer: python3 tools/copy_rate.py   (HYMLS_MI_STAGED_COPY=0 switches the staged path off)"""
import os, sys, time
os.environ["HYMLS_MI_NO_TORCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hymls_amd
lib = hymls_amd.load_library()
rp, ci, va = hymls_amd.generate_problem("Laplace", 8, 8, 8)
P = hymls_amd.Preconditioner((rp, ci, va), {"Problem": {"Equations": "Laplace", "Dimension": 3, "nx": 8, "ny": 8, "nz": 8},
                                            "Preconditioner": {"Separator Length": 4, "Number of Levels": 1}})
n = 1 << 28
a = np.arange(n, dtype=np.float64)
b = np.empty_like(a)
h = P._h if hasattr(P, "_h") else P.handle
d = lib.hymls_mi_device_alloc(h, C.c_int64(n * 8))
assert d
for rep in range(3):
    t0 = time.perf_counter(); rc = lib.hymls_mi_copy_to_device(h, C.c_void_p(d), a.ctypes.data_as(C.c_void_p), C.c_int64(n * 8)); t1 = time.perf_counter()
    assert rc == 0
    rc = lib.hymls_mi_copy_to_host(h, b.ctypes.data_as(C.c_void_p), C.c_void_p(d), C.c_int64(n * 8)); t2 = time.perf_counter()
    assert rc == 0
    print("staged=%s  H2D %.2f GB/s   D2H %.2f GB/s" % (os.environ.get("HYMLS_MI_STAGED_COPY", "1"), n * 8 / (t1 - t0) / 1e9, n * 8 / (t2 - t1) / 1e9), flush=True)
assert np.array_equal(a, b)
lib.hymls_mi_device_free(h, C.c_void_p(d))
