"""hymls_amd -- MI355X-native HYMLS preconditioner hot path.

Host-side mirror (Python, ctypes) of the reference's operator interface
``HYMLS::Preconditioner`` (an ``Ifpack_Preconditioner`` / ``Epetra_Operator``,
reference src/HYMLS_Preconditioner.hpp:56-254) on top of the C ABI declared in
``include/hymls_mi.h`` and implemented by ``hymls_amd/libhymls_mi.so`` (C++ host
code + hand-written HIP kernels for gfx950).  There is no CPU fallback: if the
HIP library is missing, importing :class:`Preconditioner` users get a loud error.
"""
from .api import (Preconditioner, HymlsError, load_library, generate_matrix, generate_testvector, generate_rows,
                  generate_testvector_rows, generate_problem, LIB_PATH)



def __getattr__(name):
    # the Krylov caller needs torch; the preconditioner itself does not (tools/pmc_driver.py runs without it)
    if name in ("Solver", "BorderedSolver"):
        from . import solver
        return getattr(solver, name)
    raise AttributeError("module 'hymls_amd' has no attribute %r" % name)

__all__ = ["Solver", "BorderedSolver", "Preconditioner", "HymlsError", "load_library", "generate_matrix", "generate_testvector", "generate_rows",
           "generate_testvector_rows", "generate_problem", "LIB_PATH"]
