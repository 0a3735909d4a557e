"""The one rule of the product's DropByValue that the reference does not have (include/hymls_mi.h:
hymls_mi_drop_by_value): a diagonal entry at rounding level, |a_ii| <= 1e-14 x the largest entry of its row, is the
structural zero it is on paper.  The oracle carries the same rule behind a switch (oracle/hymls.py:
ROUNDING_LEVEL_DIAGONAL_IS_ZERO, off = the reference's literal MatrixUtils::DropByValue, src/HYMLS_MatrixUtils.cpp:
1011-1227), so both sides are compared under the same rule here -- pattern and values bit for bit -- and the fork from
the literal rule is pinned to exactly the injected rows."""
import numpy as np
import pytest
import scipy.sparse as sp

import hymls_amd
from common import problem, xml_params, product_prec, rel_diff
from oracle import hymls as oracle_hymls
from oracle.partition import Params


def with_rounding_level_pressure_diagonals(A, scale=0.5):
    """pressure rows (var 3 of 4) get a diagonal of scale x 1e-14 x rowmax: above the absolute threshold of the literal
    rule for rowmax > 2, below the relative one"""
    A = sp.lil_matrix(A)
    rmax = np.abs(sp.csr_matrix(A)).max(axis=1).toarray().ravel()
    rows = np.arange(3, A.shape[0], 4)
    for i in rows:
        A[i, i] = scale * 1e-14 * rmax[i]
    return sp.csr_matrix(A), rows, rmax


@pytest.mark.parametrize("kind", ["RelDropDiag", "RelZeroDiag", "RelFullDiag"])
def test_drop_by_value_equals_the_oracle_under_the_same_rule(hostsim_lib, kind):
    A, _ = problem("Stokes-C", 8)
    A = (A * 64.0).tocsr()                                     # rowmax of the pressure rows well above 2
    B, rows, rmax = with_rounding_level_pressure_diagonals(A)
    assert (np.abs(B.diagonal()[rows]) > 1e-14).all()          # the literal rule keeps them ...
    R = hymls_amd.api.drop_by_value(B, 1e-14, kind, lib=hostsim_lib)
    O_same = oracle_hymls.drop_by_value(B, 1e-14, kind, rounding_level_diagonal_is_zero=True)
    O_lit = oracle_hymls.drop_by_value(B, 1e-14, kind, rounding_level_diagonal_is_zero=False)
    for M in (R, O_same, O_lit):
        M.sort_indices()
    assert np.array_equal(R.indptr, O_same.indptr) and np.array_equal(R.indices, O_same.indices)
    assert np.array_equal(R.data, O_same.data)                 # bit for bit under the same rule
    assert (O_lit.diagonal()[rows] != 0).all() and (R.diagonal()[rows] == 0).all()   # ... the product's rule does not
    # the fork is confined to the injected diagonals (and what the relative test decides with them)
    D = (O_lit - R).tocoo()
    assert set(D.row[D.data != 0]) <= set(rows)
    # without rounding-level diagonals both rules are the same function
    assert (hymls_amd.api.drop_by_value(A, 1e-14, kind, lib=hostsim_lib) != oracle_hymls.drop_by_value(A, 1e-14, kind)).nnz == 0


def value_case(lib, monkeypatch):
    """value parity under the shared rule.  Two-level Stokes 8^3 (Skew Cartesian): the retained separator pressures of K
    get a diagonal of 5e-14, which arrives in the reduced matrix as a diagonal of 5e-14 next to entries of order 100 --
    what the GPU's summation order leaves there at scale.  The literal rule keeps it, and the
    last-level solver, whose pivot-free factorisation tells pressures from velocities by their zero diagonal (reference
    src/HYMLS_MatrixUtils.cpp:1344-1352), would take those pressures for velocities; under the shared rule they are the
    structural zeros they are on paper, for the product and for the oracle alike."""
    n, sx = 8, 4
    A, tv = problem("Stokes-C", n)
    A = (A * 64.0).tocsr()
    p = Params(nx=n, ny=n, nz=n, sx=sx, levels=1, equations="Stokes-C", partitioner="Skew Cartesian").finalize()
    O0 = oracle_hymls.Preconditioner(A, p, testvector=tv).compute()
    red_gids = O0.map2[O0.schur.vsum_pos]
    pgids = red_gids[(red_gids % 4 == 3) & (O0.schur.reduced.diagonal() == 0.0)]
    assert pgids.size >= 8
    B = sp.lil_matrix(A)
    for g in pgids:
        B[g, g] = 5e-14
    B = sp.csr_matrix(B)
    lit = oracle_hymls.Preconditioner(B, p, testvector=tv).compute()
    monkeypatch.setattr(oracle_hymls, "ROUNDING_LEVEL_DIAGONAL_IS_ZERO", True)
    O = oracle_hymls.Preconditioner(B, p, testvector=tv).compute()
    sel = np.isin(red_gids, pgids)
    d_lit, d_same = lit.schur.reduced.diagonal()[sel], O.schur.reduced.diagonal()[sel]
    rmax = np.abs(lit.schur.reduced).max(axis=1).toarray().ravel()[sel]
    assert (np.abs(d_lit) > 1e-14).all() and (np.abs(d_lit) <= 1e-14 * rmax).all()     # the case the rule is about
    assert (d_same == 0.0).all()
    P = product_prec(B, tv, xml_params("Stokes-C", n, sx, 1, partitioner="Skew Cartesian"), lib)
    b = np.random.default_rng(3).uniform(-1, 1, A.shape[0])
    assert rel_diff(P.ApplyInverse(b), O.apply_inverse(b)) < 1e-9


def test_apply_inverse_with_rounding_level_pressure_diagonals(hostsim_lib, monkeypatch):
    value_case(hostsim_lib, monkeypatch)


@pytest.mark.gpu
def test_apply_inverse_with_rounding_level_pressure_diagonals_gpu(gpu_lib, monkeypatch):
    value_case(gpu_lib, monkeypatch)
