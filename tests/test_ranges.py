"""Profiler ranges (roctx on the GPU; `rocprofv3 --marker-trace`) carry the reference's timer labels,
"<class>_L<level>: <function>" with levels counted from 1 (HYMLS_LPROF, reference src/HYMLS_Macros.hpp:86-137;
labels: src/HYMLS_Preconditioner.cpp:281,402,597, src/HYMLS_MatrixBlock.cpp:77,319, src/HYMLS_SchurPreconditioner.cpp:236,286,
522,1014, src/HYMLS_CoarseSolver.cpp:133,271).  HYMLS_MI_RANGE_LOG writes them to a file; read in a child process because the
switch is looked at once per process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import hymls_amd
from common import problem, xml_params
lib = hymls_amd.load_library({lib!r})
A, tv = problem("Laplace", 16)
P = hymls_amd.Preconditioner(A, xml_params("Laplace", 16, 4, 2, 2), testVector=tv, lib=lib)
P.Initialize(); P.Compute()
x = P.ApplyInverse(np.ones(A.shape[0]))
assert np.all(np.isfinite(x))
"""


def check_log(path):
    depth, labels, stack = 0, [], []
    for line in open(path):
        if line.startswith("push "):
            stack.append(line[5:].strip())
            labels.append(stack[-1])
        else:
            assert line.strip() == "pop" and stack, "pop without push"
            stack.pop()
    assert not stack, "ranges left open: %s" % stack
    for want in ("Preconditioner_L1: Initialize", "Preconditioner_L1: Compute", "MatrixBlock_L1: Compute",
                 "SchurPreconditioner_L1: Compute", "SchurPreconditioner_L1: factor blocks", "SchurPreconditioner_L1: ComputeNextLevel",
                 "Preconditioner_L2: Initialize", "Preconditioner_L2: Compute", "CoarseSolver_L3: Compute",
                 "Preconditioner_L1: ApplyInverse", "MatrixBlock_L1: ApplyInverse", "SchurPreconditioner_L1: ApplyInverse",
                 "Preconditioner_L2: ApplyInverse", "SchurPreconditioner_L2: ApplyInverse", "CoarseSolver_L3: ApplyInverse"):
        assert want in labels, "%s missing from %s" % (want, sorted(set(labels)))
    # nesting as in the reference: the next level is computed inside ComputeNextLevel, applied inside the Schur solve
    i = labels.index("SchurPreconditioner_L1: ComputeNextLevel")
    assert labels.index("Preconditioner_L2: Compute") > i
    assert labels.count("MatrixBlock_L1: ApplyInverse") == 2      # two interior solves per ApplyInverse


def run_child(lib, tmp_path):
    log = tmp_path / "ranges.log"
    env = dict(os.environ, HYMLS_MI_RANGE_LOG=str(log))
    subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib)], check=True, env=env, timeout=600)
    check_log(log)


def test_range_labels_hostsim(hostsim_lib, tmp_path):
    from conftest import HOSTSIM
    run_child(HOSTSIM, tmp_path)


@pytest.mark.gpu
def test_range_labels_gpu(gpu_lib, tmp_path):
    run_child(None, tmp_path)      # (None: the HIP library)
