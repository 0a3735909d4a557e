"""The reference's driver flow (`hymls_main file.xml [overlay.xml ...]`, src/main.cpp:120-400) on top of
hymls_amd.Preconditioner and hymls_amd.Solver, so that the reference's own example inputs
(testSuite/*.xml, testSuite/data/*) run through this library:

    python -m hymls_amd.driver cavity.xml [overlay.xml ...]

* Teuchos XML parameter lists (nested <ParameterList>/<Parameter name type value>) incl. overlay files
  (Teuchos::updateParametersFromXmlFile, main.cpp:120-123),
* "Driver" sublist: "Read Linear System" + "Data Directory" (jac.mtx / rhs.mtx / sol.mtx, MatrixMarket,
  src/HYMLS_MainUtils.cpp:35-134) or a generated matrix ("Equations" = Laplace | Stokes-C,
  MainUtils.cpp:260-348), "Number of factorizations", "Number of solves", "RHS Available",
  "Exact Solution Available",
* "Problem", "Preconditioner", "Solver" sublists go to the operator classes unchanged.
* "Null Space Type" = "Constant P" (testSuite/cavity.xml): the constant-pressure vector becomes the border of the
  system and of the preconditioner (hymls_amd.BorderedSolver), so that "Fix Pressure Level" = false works;
  "Constant" (testSuite/integration_tests/stokes4_3D.xml): one constant per degree of freedom, scaled by
  1 / sqrt(number of cells) (create_nullspace, src/HYMLS_MainUtils.cpp:361-376); the exact solution of a generated
  right-hand side has the null space projected out (src/main.cpp:401-409).
* "x-periodic" / "y-periodic" / "z-periodic" of the "Problem" list (generated Stokes-C matrices).
Not supported: "Null Space Type" = "File" / "Checkerboard", deflation of further vectors, eigenvalue runs.
"""
import os
import re
import sys
import time
import xml.etree.ElementTree as ET

import numpy as np


def _convert(typ, val):
    typ = typ.strip().lower()
    if typ in ("int", "long", "short", "unsigned int"):
        return int(val)
    if typ in ("double", "float"):
        return float(val)
    if typ == "bool":
        return val.strip().lower() in ("1", "true")
    return val


def _read_list(node):
    out = {}
    for ch in node:
        if ch.tag == "ParameterList":
            out[ch.get("name")] = _read_list(ch)
        elif ch.tag == "Parameter":
            out[ch.get("name")] = _convert(ch.get("type", "string"), ch.get("value", ""))
    return out


def _update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _update(dst[k], v)
        else:
            dst[k] = v


def read_parameters(xml_file, *overlays):
    """Teuchos::getParametersFromXmlFile + updateParametersFromXmlFile for each overlay."""
    params = _read_list(ET.parse(xml_file).getroot())
    for f in overlays:
        _update(params, _read_list(ET.parse(f).getroot()))
    return params


def read_matrix_market(path):
    """MatrixMarket 'coordinate real general|symmetric' -> (rowptr, colind, val) CSR with sorted columns,
    or 'array real general' -> dense float64 array (vectors)."""
    with open(path) as f:
        header = f.readline().lower().split()
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        dims = [int(t) for t in line.split()]
        body = np.loadtxt(f, dtype=np.float64, ndmin=2)
    if "array" in header:
        return body.reshape(dims[1], dims[0]).T.copy() if dims[1] > 1 else body.ravel()
    n, m, nnz = dims
    r, c, v = body[:, 0].astype(np.int64) - 1, body[:, 1].astype(np.int64) - 1, body[:, 2]
    if "symmetric" in header:
        off = r != c
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, v[off]])
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    rowptr = np.zeros(n + 1, np.int32)
    np.add.at(rowptr, r + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), c.astype(np.int32), v


def run(xml_file, *overlays, lib=None, device=None, out=sys.stdout):
    """returns a dict with what the reference's driver prints: iterations, residuals, errors, times."""
    import torch
    import hymls_amd
    params = read_parameters(xml_file, *overlays)
    drv = params.pop("Driver", {})
    prob = params.setdefault("Problem", {})
    dim = prob.setdefault("Dimension", 2)
    nx = prob.setdefault("nx", 32)
    ny = prob.setdefault("ny", nx)
    nz = prob.setdefault("nz", nx if dim > 2 else 1)
    eqn = prob.get("Equations", "not-set")
    if device is None:
        device = "cuda" if torch.cuda.is_available() else None
    if device is None:
        raise RuntimeError("hymls_amd.driver needs a GPU (no CPU fallback)")
    rhs = sol = None
    if drv.get("Read Linear System", False):
        d = os.path.expandvars(drv.get("Data Directory", "not specified"))
        if drv.get("File Format", "MatrixMarket") != "MatrixMarket":
            raise ValueError("only the MatrixMarket format is supported")
        K = read_matrix_market(os.path.join(d, "jac.mtx"))
        if drv.get("RHS Available", False):
            rhs = read_matrix_market(os.path.join(d, "rhs.mtx"))
        if drv.get("Exact Solution Available", False):
            sol = read_matrix_market(os.path.join(d, "sol.mtx"))
    else:
        label = drv.get("Galeri Label", eqn)            # (main.cpp:131: the label defaults to the "Equations")
        if dim != 3 or label not in ("Laplace", "Laplace3D", "Stokes-C", "Darcy"):
            raise ValueError("generated problems: 3D Laplace, Stokes-C or Darcy")
        per = tuple(bool(prob.get("%s-periodic" % ax, False)) for ax in "xyz")
        K = hymls_amd.generate_problem({"Laplace": "Laplace", "Laplace3D": "Laplace", "Stokes-C": "Stokes", "Darcy": "Darcy"}[label],
                                       nx, ny, nz, lib=lib, periodic=per)
    n = K[0].size - 1
    rows = np.repeat(np.arange(n), np.diff(K[0]))
    tv = np.zeros(n); tv[np.unique(rows[(K[2] != 0.0) & (K[1] != rows)])] = 1.0     # create_testvector
    res = {"n": n, "solves": []}
    t0 = time.time()
    P = hymls_amd.Preconditioner(K, params, testVector=tv, lib=lib)
    P.Initialize()
    res["initialize_s"] = time.time() - t0
    null_space = drv.get("Null Space Type", "None")
    V = None
    if null_space == "Constant P":
        dof = prob.get("Degrees of Freedom", dim + 1)
        V = np.zeros((n, 1)); V[dof - 1::dof, 0] = 1.0
    elif null_space == "Constant":
        dof = prob.get("Degrees of Freedom", dim + 1 if eqn == "Stokes-C" else 1)
        V = np.zeros((n, dof))
        for d in range(dof):
            V[d::dof, d] = 1.0 / np.sqrt(n // dof)
    if V is not None:
        S = hymls_amd.BorderedSolver(P, P, params)
        S.SetBorder(V, device=device)
    elif null_space == "None":
        S = hymls_amd.Solver(P, P, params)
    else:
        raise ValueError("Null Space Type '%s' is not supported" % null_space)
    rng = np.random.default_rng(drv.get("Random Seed", 1234) if drv.get("Random Seed", -1) != -1 else 1234)
    for f in range(drv.get("Number of factorizations", 1)):
        if f > 0 and drv.get("Diagonal Perturbation", 0.0) != 0.0:
            val = K[2].copy()
            diag = K[1] == rows
            val[diag] += drv["Diagonal Perturbation"] * rng.uniform(-1, 1, int(diag.sum()))
            P.SetMatrix((K[0], K[1], val))
        t0 = time.time(); P.Compute(); res["compute_s"] = time.time() - t0
        for s in range(drv.get("Number of solves", 1)):
            x_ex = sol if sol is not None and s == 0 else None
            if rhs is not None and s == 0:
                b = torch.from_numpy(np.ascontiguousarray(rhs, dtype=np.float64)).to(device)
            else:
                if x_ex is None:
                    x_ex = rng.uniform(-1, 1, n)
                    if V is not None and null_space == "Constant":      # project the null space out (main.cpp:401-409)
                        x_ex -= V @ (V.T @ x_ex)
                b = P.MatVec(torch.from_numpy(np.ascontiguousarray(x_ex)).to(device)).clone()
            t0 = time.time()
            x = S.ApplyInverse(b)
            sb = None
            if isinstance(x, tuple):        # bordered: (X, S)
                x, sb = x
            t_solve = time.time() - t0
            rvec = b - P.MatVec(x)
            if sb is not None and len(sb):
                rvec = rvec - torch.from_numpy(V @ sb).to(rvec.device)
            r = float(rvec.norm() / b.norm())
            rec = {"iterations": S.getNumIter(), "residual": r, "solve_s": t_solve}
            if x_ex is not None:
                e = x.cpu().numpy() - x_ex
                dof = prob.get("Degrees of Freedom", dim + 1 if eqn == "Stokes-C" else 1)
                if null_space == "Constant":   # the null space is projected out of x_ex and excluded by the border
                    rec["error"] = float(np.linalg.norm(e) / np.linalg.norm(x_ex))
                elif eqn == "Stokes-C":     # the pressure is determined up to a constant: compare the velocities
                    e = e[np.arange(n) % dof != dof - 1]
                    rec["error"] = float(np.linalg.norm(e) / np.linalg.norm(x_ex[np.arange(n) % dof != dof - 1]))
                else:
                    rec["error"] = float(np.linalg.norm(e) / np.linalg.norm(x_ex))
            res["solves"].append(rec)
            print("solve %d.%d: %d iterations, residual %.3e%s, %.3f s" %
                  (f, s, rec["iterations"], r, ", error %.3e" % rec["error"] if "error" in rec else "", t_solve), file=out)
    res["levels"] = P.level_sizes()
    if res["solves"]:                      # what the reference's "Targets" list is checked against
        last = res["solves"][-1]
        res["iterations"], res["relative_residual"], res["relative_error"] = last["iterations"], last["residual"], last.get("error")
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print("USAGE: python -m hymls_amd.driver <parameter_filename> [overlay.xml ...]")
        return 0
    run(argv[0], *argv[1:])
    return 0


if __name__ == "__main__":
    sys.exit(main())
