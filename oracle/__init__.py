"""CPU restatement ("oracle") of the HYMLS preconditioner setup+apply hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product
(``hymls_amd``) never imports this package and fails loudly when its HIP
library is missing.

Each function cites the reference file:line (relative to ``/root/reference``)
whose behaviour it restates.  The numerics are plain numpy/scipy (FP64); the
sparse/dense LU factorisations use scipy (SuperLU / LAPACK), which the
reference's own tests do not pin either (any backward-stable LU is accepted:
SURVEY.md section 8c).

Parity pin status (see tests/test_oracle_*.py):
  * generators: pinned bit-for-bit against the reference fixture
    testSuite/data/DrivenCavity/16x16x16/Re0/jac.mtx (committed as a checksum
    plus an 4^3 slice under tests/golden/),
  * partitioner: pinned by the group-count / interior-size formulas and node
    lists of testSuite/unit_tests/HYMLS_OverlappingPartitioner.cpp:220-672,
  * preconditioner: pinned by the exactness tests (levels=0 => exact inverse,
    testSuite/unit_tests/HYMLS_Preconditioner.cpp:247-276) and the
    iteration-count targets of testSuite/integration_tests/threeD1.xml.
The reference itself cannot be compiled here (Trilinos absent), so LU *values*
are "parity unpinned" by construction, exactly as in the reference's tests.
"""
