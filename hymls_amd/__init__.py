"""hymls_amd -- MI355X-native HYMLS preconditioner hot path.

Host-side mirror (Python, ctypes) of the reference's operator interface
``HYMLS::Preconditioner`` (an ``Ifpack_Preconditioner`` / ``Epetra_Operator``,
reference src/HYMLS_Preconditioner.hpp:56-254) on top of the C ABI declared in
``include/hymls_mi.h`` and implemented by ``hymls_amd/libhymls_mi.so`` (C++ host
code + hand-written HIP kernels for gfx950).  There is no CPU fallback: if the
HIP library is missing, importing :class:`Preconditioner` users get a loud error.
"""
from .api import (Preconditioner, HymlsError, load_library, generate_matrix, generate_testvector, generate_rows,
                  generate_testvector_rows, LIB_PATH)

from .solver import Solver, BorderedSolver

__all__ = ["Solver", "BorderedSolver", "Preconditioner", "HymlsError", "load_library", "generate_matrix", "generate_testvector", "generate_rows",
           "generate_testvector_rows", "LIB_PATH"]
