"""Worker of the sharded-path tests (launched through torch.distributed.run, one process per rank):
builds the sharded preconditioner, applies it to this rank's part of a seeded global vector and lets
rank 0 compare the assembled result with the one-rank preconditioner on the same problem.
  python -m torch.distributed.run --nproc-per-node W tests/dist_worker.py EQ NX NY NZ SX LEVELS CX PART MODE
MODE = hostsim (TEST-ONLY CPU simulator of the device plan, gloo) | gpu (all ranks share cuda:0, gloo staging)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import hymls_amd
from hymls_amd.dist import TorchComm, rank_grid


def main():
    eq, nx, ny, nz, sx, levels, cx, part, mode = sys.argv[1:10]
    nx, ny, nz, sx, levels, cx = int(nx), int(ny), int(nz), int(sx), int(levels), int(cx)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if mode == "gpu-rccl":     # built-in transport of the library (comm_rccl.cpp); torch.distributed (gloo) only hands out the id
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend="gloo")
    elif mode == "gpu-nccl":   # one rank per GPU over RCCL (a single rank exchanges with itself: HYMLS_MI_FORCE_SHARDED)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode == "gpu-nccl":
        lib = hymls_amd.load_library()
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
        from hymls_amd.dist import transport_selftest
        err = transport_selftest(device, "nccl")
        assert err is None, err
    elif mode == "gpu-rccl":
        lib = hymls_amd.load_library()
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    elif mode == "hostsim":
        lib = hymls_amd.load_library(os.path.join(ROOT, "tests", "hostsim", "libhymls_mi_hostsim.so"))
        device = "cpu"
    else:
        lib = hymls_amd.load_library()
        device = "cuda:0"
    prec = {"Separator Length": sx, "Number of Levels": levels, "Partitioner": part}
    if cx > 0:
        prec["Coarsening Factor"] = cx
    prm = {"Problem": {"Equations": eq, "Dimension": 3, "nx": nx, "ny": ny, "nz": nz}, "Preconditioner": prec}
    # HYMLS_TEST_PERIODIC=xyz letters: periodic directions (GaleriExt periodic matrices, wrap-around partitioner)
    per = tuple(ax in os.environ.get("HYMLS_TEST_PERIODIC", "") for ax in "xyz")
    for ax, on in zip("xyz", per):
        if on:
            prm["Problem"]["%s-periodic" % ax] = True
    a = float(nx * nx)
    if mode == "gpu-rccl":
        from hymls_amd.dist import RcclComm
        comm = RcclComm(int(os.environ.get("LOCAL_RANK", "0")), lib=lib)
    else:
        comm = TorchComm(device)
    P = hymls_amd.Preconditioner(None, prm, lib=lib, comm=comm, rank_grid=rank_grid(world),
                                 device=int(os.environ.get("LOCAL_RANK", "0")) if mode in ("gpu-nccl", "gpu-rccl") else 0)
    assert P.CommSelfTest() == 0, "transport self-test failed"
    req = P.RequiredRows()
    if any(per):
        rows = hymls_amd.generate_problem("Stokes" if eq != "Laplace" else "Laplace", nx, ny, nz, a=a, gids=req, lib=lib, periodic=per)
    else:
        rows = hymls_amd.generate_rows(eq, nx, ny, nz, req, a=a, lib=lib)
    P.SetMatrixRows(req, rows)
    P.SetTestVector(hymls_amd.generate_testvector_rows(req, *rows))
    P.Initialize()
    P.Compute()
    owned = P.OwnedRows()
    N = nx * ny * nz * (1 if eq == "Laplace" else 4)
    b = np.random.default_rng(5).uniform(-1, 1, N)
    x_loc = P.ApplyInverse(b[owned])
    x_loc2 = P.ApplyInverse(b[owned])          # a second call must reproduce the first (buffers reused)
    P.SetMatrixRows(req, rows)                 # SetMatrix with the same pattern + Compute again
    P.Compute()
    x_loc3 = P.ApplyInverse(b[owned])
    # sharded K x and the sharded Krylov loop (hostsim runs: CPU tensors)
    kx = P.MatVec(b[owned])
    its_sh, xs_loc = -1, None
    if mode == "hostsim":
        meth = "CG" if eq == "Laplace" else "GMRES"
        S = hymls_amd.Solver(P, P, {"Krylov Method": meth, "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 300, "Num Blocks": 300}})
        x_ex = np.random.default_rng(9).uniform(-1, 1, N)          # consistent right-hand side b = K x_ex
        rhs_loc = P.MatVec(x_ex[owned])
        xs_loc = S.ApplyInverse(torch.from_numpy(rhs_loc.copy())).numpy()
        its_sh = S.getNumIter()
    # bordered system on the sharded handle (HYMLS_TEST_BORDER=1): every rank passes its rows of V and W
    xb_loc, sb = None, None
    if os.environ.get("HYMLS_TEST_BORDER"):
        rngb = np.random.default_rng(17)
        Vg, Wg, Cg = rngb.uniform(-1, 1, (N, 2)), rngb.uniform(-1, 1, (N, 2)), rngb.uniform(-1, 1, (2, 2))
        tb = rngb.uniform(-1, 1, 2)
        P.SetBorder(Vg[owned], Wg[owned], Cg)
        P.Compute()
        xb_loc, sb = P.ApplyInverseBordered(b[owned], tb)
        P.SetBorder(None)
        P.Compute()
    parts = [None] * world
    dist.all_gather_object(parts, (owned, x_loc, float(np.abs(x_loc2 - x_loc).max()), float(np.abs(x_loc3 - x_loc).max()), kx, xs_loc, xb_loc, sb))
    ok = True
    if rank == 0:
        x = np.full(N, np.nan)
        cover = np.zeros(N, np.int64)
        kxg = np.zeros(N); xsg = np.zeros(N)
        xbg = np.zeros(N)
        for o, xl, _, _, kxl, xsl, xbl, _sb in parts:
            x[o] = xl
            cover[o] += 1
            kxg[o] = kxl
            if xsl is not None:
                xsg[o] = xsl
            if xbl is not None:
                xbg[o] = xbl
        if any(per):
            K = hymls_amd.generate_problem("Stokes" if eq != "Laplace" else "Laplace", nx, ny, nz, a=a, lib=lib, periodic=per)
        else:
            K = hymls_amd.generate_matrix(eq, nx, ny, nz, a=a, lib=lib)
        tv = hymls_amd.generate_testvector(*K, lib=lib)
        P0 = hymls_amd.Preconditioner(K, prm, testVector=tv, lib=lib)
        P0.Compute()
        x0 = P0.ApplyInverse(b)
        err = float(np.linalg.norm(x - x0) / np.linalg.norm(x0))
        import scipy.sparse as sp
        Ks = sp.csr_matrix((K[2], K[1], K[0]), shape=(N, N))
        mv_err = float(np.linalg.norm(kxg - Ks @ b) / np.linalg.norm(Ks @ b))
        its_one, sol_res = -1, 0.0
        if its_sh >= 0:
            S0 = hymls_amd.Solver(P0, P0, {"Krylov Method": meth, "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 300, "Num Blocks": 300}})
            rhs = Ks @ np.random.default_rng(9).uniform(-1, 1, N)
            S0.ApplyInverse(torch.from_numpy(rhs.copy()))
            its_one = S0.getNumIter()
            sol_res = float(np.linalg.norm(rhs - Ks @ xsg) / np.linalg.norm(rhs))
        border_err = 0.0
        if parts[0][6] is not None:
            P0.SetBorder(Vg, Wg, Cg)
            P0.Compute()
            xb0, sb0 = P0.ApplyInverseBordered(b, tb)
            border_err = max(float(np.linalg.norm(xbg - xb0) / np.linalg.norm(xb0)),
                             max(float(np.abs(p[7] - sb0).max() / max(np.abs(sb0).max(), 1e-300)) for p in parts))
            P0.SetBorder(None)
            P0.Compute()
        res = {"world": world, "cover_ok": bool((cover == 1).all()), "rel_err": err, "matvec_err": mv_err, "border_err": border_err,
               "krylov_its_sharded": its_sh, "krylov_its_one_rank": its_one, "krylov_residual": sol_res,
               "repeat_diff": max(p[2] for p in parts), "recompute_diff": max(p[3] for p in parts),
               "levels": P0.level_sizes(), "levels_sharded": P.level_sizes()}
        print("DIST_RESULT " + json.dumps(res), flush=True)
        ok = (res["cover_ok"] and err < 1e-9 and res["repeat_diff"] == 0.0 and res["recompute_diff"] < 1e-12 and mv_err < 1e-13
              and abs(its_sh - its_one) <= 1 and sol_res < 1e-6 and border_err < 1e-9)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
