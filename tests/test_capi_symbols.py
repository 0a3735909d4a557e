"""-m "not gpu": the C-ABI library loads and exports every symbol include/hymls_mi.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import hymls_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hymls_mi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hymls_mi_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_by_product_library():
    syms = declared_symbols()
    assert len(syms) >= 30
    assert os.path.exists(hymls_amd.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(hymls_amd.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libhymls_mi.so does not export %s" % s


def test_python_binding_covers_header():
    lib = hymls_amd.load_library()
    assert set(declared_symbols()) == set(lib._hymls_symbols)


def test_product_library_contains_gfx950_code_object():
    data = open(hymls_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in data, "no gfx950 code object embedded"
    for k in (b"k_solve_fwd", b"k_solve_bwd", b"k_factor_level", b"k_spmv", b"k_ot", b"k_blocks_apply"):
        assert k in data


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU the product must fail loudly (code -3), not fall back."""
    import torch
    if torch.cuda.is_available():
        return
    import numpy as np
    from common import problem, xml_params
    A, tv = problem("Laplace", 8)
    try:
        hymls_amd.Preconditioner(A, xml_params("Laplace", 8, 4, 0), testVector=tv)
    except hymls_amd.HymlsError as e:
        assert e.code == -3 and "no HIP device" in str(e)
    else:
        raise AssertionError("creating a preconditioner without a GPU must fail")
