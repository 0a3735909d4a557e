"""Generates the committed golden fixtures from the reference's own test data.
Run once in the dev container (needs /root/reference):  python tests/golden/make_golden.py

* stokes3d_16_fixture.json : sha256 of the canonical CSR arrays of
  testSuite/data/DrivenCavity/16x16x16/Re0/jac.mtx with the pressure COLUMNS negated
  (what the reference's own test compares GaleriExt::Stokes3D(16^3, a=dx, b=dx^2) with,
  testSuite/unit_tests/GaleriExt_Stokes3D.cpp:19-62) -- the 4 MB matrix itself is not committed.
* drivencavity16_rhs_sol.npz : rhs.mtx and sol.mtx of the same directory (data files of the
  reference's integration tests stokes{0,1,2}_3D.xml), stored as float64 arrays.
* stokes3d_4.npz : the restated generator's 4^3 matrix (a=1/4, b=1/16) after it reproduced the
  fixture bit for bit, as a small known-answer vector for the C++ generator.
* drivencavity32_2d_re{0,1000}.npz, drivencavity64_2d_re1000.npz : jac.mtx / rhs.mtx / sol.mtx of
  testSuite/data/DrivenCavity/{32x32/Re0, 32x32/Re1000, 64x64/Re1000}
  (the input of testSuite/cavity.xml): a 2D Navier-Stokes Jacobian (3 dof per cell, nonsymmetric at Re 1000)
  with its right-hand side and solution, as CSR arrays.
"""
import hashlib
import json
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import galeri  # noqa: E402

D = "/root/reference/testSuite/data/DrivenCavity/16x16x16/Re0/"


def canonical_sha(A):
    A = A.tocsr().copy()
    A.sum_duplicates()
    A.sort_indices()
    h = hashlib.sha256()
    h.update(np.asarray(A.indptr, dtype=np.int64).tobytes())
    h.update(np.asarray(A.indices, dtype=np.int64).tobytes())
    h.update(np.asarray(A.data, dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    F = scipy.io.mmread(D + "jac.mtx").tocsr()
    s = np.ones(F.shape[0]); s[3::4] = -1.0
    F = (F @ sp.diags(s)).tocsr()
    G = galeri.stokes3d(16, 16, 16, a=1.0 / 16, b=1.0 / 256)
    assert canonical_sha(F) == canonical_sha(G), "restated generator does not reproduce the reference fixture"
    json.dump({"source": "testSuite/data/DrivenCavity/16x16x16/Re0/jac.mtx (pressure columns negated)",
               "shape": list(F.shape), "nnz": int(F.nnz), "sha256": canonical_sha(F)},
              open(os.path.join(HERE, "stokes3d_16_fixture.json"), "w"), indent=1)
    rhs = np.asarray(scipy.io.mmread(D + "rhs.mtx")).ravel()
    sol = np.asarray(scipy.io.mmread(D + "sol.mtx")).ravel()
    np.savez_compressed(os.path.join(HERE, "drivencavity16_rhs_sol.npz"), rhs=rhs, sol=sol)
    A4 = galeri.stokes3d(4, 4, 4, a=0.25, b=1.0 / 16)
    np.savez_compressed(os.path.join(HERE, "stokes3d_4.npz"), indptr=A4.indptr, indices=A4.indices, data=A4.data)
    for grid, re in ((32, "Re0"), (32, "Re1000"), (64, "Re1000")):
        d2 = "/root/reference/testSuite/data/DrivenCavity/%dx%d/%s/" % (grid, grid, re)
        J = scipy.io.mmread(d2 + "jac.mtx").tocsr()
        J.sum_duplicates(); J.sort_indices()
        np.savez_compressed(os.path.join(HERE, "drivencavity%d_2d_%s.npz" % (grid, re.lower())),
                            indptr=J.indptr.astype(np.int32), indices=J.indices.astype(np.int32), data=J.data,
                            rhs=np.asarray(scipy.io.mmread(d2 + "rhs.mtx")).ravel(),
                            sol=np.asarray(scipy.io.mmread(d2 + "sol.mtx")).ravel())
    print("golden fixtures written")


if __name__ == "__main__":
    main()
