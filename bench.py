#!/usr/bin/env python
"""bench.py -- Preconditioner ApplyInverse throughput (DoF/s) + achieved HBM GB/s.

Contract (see task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  A "step" is one ApplyInverse (one right-hand side, device-resident vectors) of
the HYMLS preconditioner computed for the synthetic GaleriExt Stokes3D Jacobian the metric of
BASELINE.json is quoted on: Stokes3D 256^3 (67 108 864 DoF), 3-level (XML "Number of Levels" = 2),
separator length 8, coarsening factor 8 (configs[2]); it fits one MI355X (about 60 GB of factors).
The 3D Stokes-C problem is partitioned with the reference's "Skew Cartesian" partitioner, the only
one the reference itself can run 3D Stokes with.  `--grid 128 --levels 1` is configs[1].

N > 1 (launched through torch.distributed.run, one rank per GPU): the SAME 256^3 problem is sharded
(strong scaling, as the metric says "256^3 Stokes3D, 1/2/4/8 GPU"): rank r owns one box of the grid
(2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2 boxes), the subdomains in it and the separators they list first;
halo values and the V-sum hand-off travel through torch.distributed (RCCL) -- DESIGN.md section
"multi-GPU".  value = global DoF / time.  `--replicas` runs N independent copies instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def cpu_baseline(problem, re, n, sx, levels, seconds_budget=30.0, gpu_lib_ok=True):
    """The compiled CPU oracle (oracle/cpu/hymls_cpu.cpp through oracle/cpu_oracle.py: the reference's algorithm as the
    reference performs it -- per-subdomain sparse LU with the F-matrix ordering, dense Schur parts, Householder, dgetrf
    blocks, serial exact coarse LU -- OpenMP over subdomains, built here with -O3 -march=native; kind 'port') timed on
    the host cores of this box on a bounded sample of the workload: the same problem family, partitioner and separator
    length on an n^3 grid, ApplyInverse only, at 1 thread (= one rank of the reference) and at all cores."""
    import tempfile
    from oracle import galeri, cpu_oracle
    from oracle.partition import Params
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the GPU box grants 16 host cores per GPU whatever the affinity mask says (and 256 threads on a 0.4 M DoF sample
    # are slower than one: measured 0.41 against 8.0 MDoF/s)
    cores = max(1, min(cores, int(os.environ.get("HYMLS_CPU_BASELINE_THREADS", "16"))))
    try:
        lib = cpu_oracle.load(cpu_oracle.build(native=True, out_dir=tempfile.mkdtemp(prefix="hymls_cpu_")))
    except Exception:  # pragma: no cover  (no compiler on the box: the portable build made by __graft_entry__.build())
        lib = cpu_oracle.load()
    A = {"stokes": lambda: galeri.stokes3d(n, n, n), "darcy": lambda: galeri.darcy3d(n, n, n, 1.0, -1.0),
         "cavity": lambda: galeri.oseen3d(n, n, n, re)}[problem]()
    tv = galeri.create_testvector(A)
    p = Params(nx=n, ny=n, nz=n, sx=sx, levels=levels, equations="Stokes-C", partitioner="Skew Cartesian").finalize()
    t0 = time.time()
    O = cpu_oracle.Preconditioner(A, p, testvector=tv, nthreads=cores, lib=lib).compute()
    t_setup = time.time() - t0
    b = np.random.default_rng(0).uniform(-1, 1, A.shape[0])
    rate = {}
    for nt in (1, cores):
        O.set_threads(nt)
        O.apply_inverse(b)
        reps, t0 = 0, time.time()
        while reps < 3 or (time.time() - t0 < min(5.0, seconds_budget / 4) and reps < 50):
            O.apply_inverse(b)
            reps += 1
        rate[nt] = (A.shape[0] / ((time.time() - t0) / reps), reps)
    # Krylov iteration count on the sample ("must equal the CPU path's on the same matrix"): right-preconditioned
    # GMRES, zero initial guess, b = K x_ex, 1e-8 -- with the CPU oracle here, and with the product on the SAME sample
    # (same matrix, same right-hand side) on the GPU
    from oracle import krylov
    x_ex = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    rhs = A @ x_ex
    O.set_threads(cores)
    _, its_cpu, res_cpu = krylov.gmres(lambda v: A @ v, rhs, O.apply_inverse, tol=1e-8, maxit=300)
    its_gpu, res_gpu = None, None
    if gpu_lib_ok:
        import hymls_amd
        prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
               "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
        A.sort_indices()
        Pg = hymls_amd.Preconditioner(A, prm, testVector=tv)
        Pg.Compute()
        Sg = hymls_amd.Solver(Pg, Pg, {"Krylov Method": "GMRES", "Iterative Solver": {"Convergence Tolerance": 1e-8, "Maximum Iterations": 300, "Num Blocks": 300}})
        xg = Sg.ApplyInverse(torch.from_numpy(rhs).cuda())
        its_gpu = Sg.getNumIter()
        res_gpu = float(np.linalg.norm(rhs - A @ xg.cpu().numpy()) / np.linalg.norm(rhs))
        del Sg, Pg
    return {"value": rate[cores][0], "unit": "DoF/s", "cores": cores, "kind": "port",
            "value_1_core": rate[1][0],
            "krylov_iterations_cpu": its_cpu, "krylov_iterations_gpu_same_sample": its_gpu,
            "krylov_true_relative_residual_cpu": res_cpu, "krylov_true_relative_residual_gpu_same_sample": res_gpu,
            "sample": "compiled CPU oracle (g++ -O3 -march=native -fopenmp; sparse LU per subdomain, %d nonzeros in L+U; serial SuperLU coarse "
                      "solve) ApplyInverse on %s %d^3 = %d DoF, Skew Cartesian sx=%d, Number of Levels=%d, levels %s; %d applies at %d "
                      "threads, %d at 1 thread, after a %.1f s setup on %d threads; right-preconditioned GMRES to 1e-8 on this sample: "
                      "%d iterations with the CPU oracle, %s with the product on the GPU (same matrix, same right-hand side)"
                      % (O.nnz_factors(), problem, n, A.shape[0], sx, levels, O.level_sizes(), rate[cores][1], cores, rate[1][1], t_setup, cores,
                         its_cpu, its_gpu)}


PROBLEM = {"stokes": "Stokes", "darcy": "Darcy", "cavity": "Cavity"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="global grid size per direction")
    ap.add_argument("--sx", type=int, default=8)
    ap.add_argument("--levels", type=int, default=2, help="XML 'Number of Levels' (2 = the 3-level method)")
    ap.add_argument("--problem", choices=["stokes", "darcy", "cavity"], default="stokes",
                    help="input matrix: GaleriExt Stokes3D (configs[1], [2]), GaleriExt Darcy3D (configs[4]), "
                         "Navier-Stokes-like Jacobian at --re (configs[3]; synthesised, see oracle/galeri.py:oseen3d)")
    ap.add_argument("--re", type=float, default=1000.0, help="Reynolds number of --problem cavity")
    ap.add_argument("--nvec", type=int, default=1,
                    help="right-hand sides per ApplyInverse (Epetra_MultiVector columns); > 1: the factors are streamed once per "
                         "group of 4 columns, value = DoF x vectors / s")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent copies instead of the sharded problem")
    ap.add_argument("--krylov", dest="krylov", action="store_true", default=True,
                    help="(default) after the timed region: solve K x = b (b = K x_ex) with right-preconditioned GMRES on the "
                         "device (hymls_amd.Solver, relative residual 1e-8) and report the iteration count and time")
    ap.add_argument("--no-krylov", dest="krylov", action="store_false")
    ap.add_argument("--krylov-restart", type=int, default=100, help="GMRES restart length ('Num Blocks')")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL ONLY: all ranks use cuda:0 (with --backend gloo, staged through the host); "
                         "the numbers are not a multi-GPU measurement and are marked so")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N = 1: take the sharded code path anyway (one rank exchanging with itself over the transport): "
                         "measures what packing + callbacks + collectives cost per ApplyInverse")
    ap.add_argument("--cpu-n", type=int, default=48, help="grid size of the CPU baseline's bounded sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="sharded runs: 'rccl' = the library's built-in transport (ncclSend/ncclRecv groups on its stream, "
                         "no Python inside ApplyInverse); 'torch' = callbacks into torch.distributed.all_to_all_single")
    ap.add_argument("--hostsim", action="store_true",
                    help="TEST ONLY: drive the host-logic simulator on the CPU (tests/test_bench_multirank.py); "
                         "numbers produced this way are meaningless and are marked as such")
    args = ap.parse_args()
    # exactly ONE line on stdout: libraries (RCCL prints its version banner there) write to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU now, before anything touches the GPU
        # (never re-launch from a process that has initialised HIP), and leave with the launcher's exit code
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, stdout=json_fd))   # the children's one JSON line goes to the real stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world and not args.hostsim:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-GPU number as the %d-GPU point\n"
                         % (args.gpus, world, world, args.gpus))
        sys.exit(2)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import hymls_amd
    if args.hostsim:
        lib = hymls_amd.load_library(os.path.join(ROOT, "tests", "hostsim", "libhymls_mi_hostsim.so"))
        dev = torch.device("cpu")
        backend = "gloo"
    else:
        lib = None
        backend = args.backend
        assert torch.cuda.is_available(), "bench.py needs a GPU (hymls_amd has no CPU fallback)"
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if args.force_sharded and world == 1:
        os.environ["HYMLS_MI_FORCE_SHARDED"] = "1"
        os.environ.setdefault("MASTER_PORT", "29577"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    n, sx = args.n, args.sx
    sharded = (world > 1 and not args.replicas) or args.force_sharded
    note = None
    if sharded:
        # self-test of the transport on this backend (uneven all-to-all on the library's stream); a failure ends the
        # run with a non-zero exit code: N replicas under the sharded metric would inflate the number
        from hymls_amd.dist import TorchComm, RcclComm, rank_grid, transport_selftest
        native = args.transport == "rccl" and not args.hostsim and not args.share_gpu and backend == "nccl"
        err = None if native else transport_selftest(dev, backend)   # (the built-in transport fails loudly by itself)
        if err:
            sys.stderr.write("bench.py: sharded transport self-test failed on rank %d (%s); no fallback\n" % (rank, err))
            dist.destroy_process_group()
            sys.exit(3)
    levels = args.levels
    if sharded:
        px, py, pz = rank_grid(world)
        nx, ny, nz = n, n, n
        prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": nx, "ny": ny, "nz": nz},
               "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
        comm = RcclComm(local_rank, lib=lib) if native else TorchComm(dev)
        P = hymls_amd.Preconditioner(None, prm, device=local_rank, lib=lib, comm=comm, rank_grid=(px, py, pz))
        if native:
            # the built-in transport checks itself on every rank before any work is sharded over it.  A failure on any rank
            # ends the run with a non-zero exit code: the callback transport is a different code path and has to be asked
            # for (--transport torch), never switched to behind the reader's back
            self_rc = P.CommSelfTest()
            bad = torch.tensor([1.0 if self_rc != 0 else 0.0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bad.item() > 0:
                sys.stderr.write("bench.py: the built-in RCCL transport failed its self-test (rank %d: %s); no fallback -- "
                                 "rerun with --transport torch to use the torch.distributed callbacks\n"
                                 % (rank, "failed here with code %d" % self_rc if self_rc != 0 else "ok here, failed on another rank"))
                del P
                comm.close()
                dist.destroy_process_group()
                sys.exit(3)
        t0 = time.time()
        req = P.RequiredRows()
        rows = hymls_amd.generate_problem(PROBLEM[args.problem], nx, ny, nz, re=args.re, gids=req, lib=lib)
        P.SetMatrixRows(req, rows)
        P.SetTestVector(hymls_amd.generate_testvector_rows(req, *rows))
        del rows
        P.Initialize(); t_init = time.time() - t0
        t0 = time.time(); P.Compute(); t_comp = time.time() - t0
        t0 = time.time(); P.Compute(); t_recomp = time.time() - t0     # SetMatrix-style recompute: symbolic work reused
        N_local = P.OwnedRows().size
        N_global = nx * ny * nz * 4
    else:
        nx = ny = nz = n
        rp, ci, va = hymls_amd.generate_problem(PROBLEM[args.problem], n, n, n, re=args.re, lib=lib)
        tv = hymls_amd.generate_testvector(rp, ci, va, lib=lib)
        prm = {"Problem": {"Equations": "Stokes-C", "Dimension": 3, "nx": n, "ny": n, "nz": n},
               "Preconditioner": {"Separator Length": sx, "Number of Levels": levels, "Partitioner": "Skew Cartesian"}}
        P = hymls_amd.Preconditioner((rp, ci, va), prm, testVector=tv, device=local_rank, lib=lib)
        t0 = time.time(); P.Initialize(); t_init = time.time() - t0
        t0 = time.time(); P.Compute(); t_comp = time.time() - t0
        t0 = time.time(); P.Compute(); t_recomp = time.time() - t0     # SetMatrix-style recompute: symbolic work reused
        N_local = rp.size - 1
        N_global = N_local * world
        del rp, ci, va
    N = N_local
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    b = torch.rand((args.nvec, N) if args.nvec > 1 else N, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    x = torch.empty_like(b)
    if args.hostsim:   # the simulator works on host memory: hand it numpy views
        b, x = b.numpy(), x.numpy()

    def barrier():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        P.ApplyInverse(b, x)
    P.set_profiling(True)   # hipEvents on the library's stream, no synchronisation inside the region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        P.ApplyInverse(b, x)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    t_phase = [P.last_apply_seconds(i) for i in range(5)]   # averages over the timed steps
    P.set_profiling(False)
    assert bool(np.isfinite(x).all() if args.hostsim else torch.isfinite(x).all()), "ApplyInverse produced non-finite values"
    bytes_all = [P.apply_bytes(i) for i in range(9)]
    bytes_rank0 = list(bytes_all)
    flops_all = [P.setup_flops(i) for i in range(4)]
    if world > 1:   # algorithmic bytes / flops of the whole job = sum over the ranks
        t = torch.tensor(bytes_all + flops_all, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bytes_all, flops_all = [float(v) for v in t.tolist()[:9]], [float(v) for v in t.tolist()[9:]]

    hbm_used = None
    if not args.hostsim:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        hbm_used = (total_b - free_b) / 2**30
    krylov = None
    if args.krylov:
        S = hymls_amd.Solver(P, P, {"Krylov Method": "GMRES", "Iterative Solver": {
            "Convergence Tolerance": 1e-8, "Maximum Iterations": 2000, "Num Blocks": args.krylov_restart, "Maximum Restarts": 40}})
        if args.hostsim:
            bt = torch.from_numpy(b)
        else:
            bt = b
        if bt.dim() > 1:
            bt = bt[0]          # (--nvec > 1: the Krylov check uses the first column)
        rhs = P.MatVec(bt).clone()
        barrier(); t0 = time.perf_counter()
        xs = S.ApplyInverse(rhs)
        barrier(); t_k = time.perf_counter() - t0
        r = rhs - P.MatVec(xs)
        rr = torch.stack([torch.dot(r, r), torch.dot(rhs, rhs)])
        if world > 1:
            dist.all_reduce(rr)
        krylov = {"method": "GMRES(%d), right preconditioned, zero initial guess, b = K x_ex" % args.krylov_restart,
                  "tolerance": 1e-8, "iterations": S.getNumIter(), "seconds": t_k,
                  "true_relative_residual": float((rr[0] / rr[1]).sqrt())}
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        # dominant kernel: k_interior_fused (two launches per ApplyInverse).  Algorithmic bytes
        # per launch = every stored factor-panel entry once (8 B, forward reads L-side, backward
        # U-side) + the interior vector in and out.
        lv = P.level_sizes()
        n1 = (lv[0][1] - lv[0][2]) / (world if sharded else 1)   # interior unknowns per GPU
        # (the smaller of the stored footprint and the sparse-equivalent nnz(L+U) x 12 B + 24 B per unknown)
        bytes_launch = min(bytes_rank0[1], bytes_rank0[6]) / 2.0 + 16.0 * n1          # this rank's launch
        t_launch = t_phase[1] / 2.0
        achieved = bytes_launch / t_launch / 1e9 if t_launch > 0 else None
        # HBM traffic of that kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
        # gfx950 1/2-fetch correction checked in the same pass on k_axpby; tools/pmc_driver.py + tools/pmc_summary.py):
        # committed measurement under profiles/, used only when it was taken on this very workload.
        traffic, traffic_other = None, None
        import glob
        for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_interior_fused.json")), reverse=True):
            try:
                pj = json.load(open(prof))
                if (pj.get("n") == n and pj.get("sx") == sx and pj.get("levels") == levels and args.problem == "stokes"
                        and world == 1):
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_other = {"source": os.path.basename(prof), "traffic_over_algorithmic": pj.get("traffic_over_algorithmic")}
                    break
            except Exception:
                pass
        # one step with nvec columns: factors, separator blocks and coarser levels once per group of 4 columns,
        # A12 / A21 and the vectors once per column
        groups = -(-args.nvec // 4)
        bytes_step = ((min(bytes_all[1], bytes_all[6]) + min(bytes_all[4], bytes_all[7]) + bytes_all[3]) * groups
                      + (bytes_all[2] + bytes_all[5]) * args.nvec)
        out = {
            "metric": "Preconditioner ApplyInverse DoF/s + achieved HBM GB/s, %s" % {"stokes": "Stokes3D", "darcy": "Darcy3D", "cavity": "cavity3D Re=%g" % args.re}[args.problem],
            "value": N_global * args.nvec * args.steps / elapsed, "unit": "DoF/s" if args.nvec == 1 else "DoF*vectors/s",
            "nvec": args.nvec, "ms_per_vector": ms / args.nvec,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong" if sharded or world == 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": ("synthetic (HOST SIMULATOR, TEST ONLY - not a measurement)" if args.hostsim else
                     "synthetic (REHEARSAL: all ranks share one GPU - not a multi-GPU measurement)" if args.share_gpu else "synthetic"),
            "config": {"workload": "%s %dx%dx%d, %d DoF, HYMLS %d-level "
                                   "(Number of Levels=%d), Skew Cartesian sx=%d, Block Diagonal, %d rhs"
                                   % ({"stokes": "GaleriExt Stokes3D (a=nx^2,b=1)", "darcy": "GaleriExt Darcy3D (a=1,b=-1)",
                                       "cavity": "Navier-Stokes-like Jacobian Re=%g (Stokes3D + central convection, synthesised)" % args.re}[args.problem],
                                      nx, ny, nz, N_global if sharded else N_local, levels + 1, levels, sx, args.nvec),
                       "parallelism": "1 GPU" if world == 1 and not sharded else (
                           "sharded: %dx%dx%d boxes of %dx%dx%d cells, one per GPU; halo + V-sum exchange over %s, ranks = %d"
                           % (px, py, pz, nx // px, ny // py, nz // pz,
                              "the built-in RCCL transport (ncclSend/ncclRecv groups on the library's stream)" if native
                              else "torch.distributed (%s) callbacks" % backend, world) if sharded else
                           (note or "%d replicas (one problem per GPU, no exchange)" % world)),
                       "levels": lv, "initialize_s": t_init, "compute_s": t_comp, "recompute_s": t_recomp,
                       "hbm_used_gib_rank0": hbm_used},
            "hbm_gbps": bytes_step / (elapsed / args.steps) / 1e9,
            "bytes_per_vector": bytes_step / args.nvec,
            "apply_bytes": {"total": bytes_all[8], "total_as_stored": bytes_all[0],
                            "interior_factors_stored": bytes_all[1], "interior_factors_sparse_equivalent": bytes_all[6],
                            "a12_a21": bytes_all[2], "separator_blocks_ot": bytes_all[3],
                            "coarse_stored": bytes_all[4], "coarse_sparse_equivalent": bytes_all[7], "vectors": bytes_all[5],
                            "note": "total = the smaller of stored (8 B x dense supernodal panel entries) and sparse-equivalent "
                                    "(12 B x nnz(L+U) + 24 B per unknown) for every factor"},
            "phase_ms": {"apply": 1e3 * t_phase[0], "interior_solves(2 launches)": 1e3 * t_phase[1], "spmv": 1e3 * t_phase[2],
                         "schur": 1e3 * t_phase[3], "coarse": 1e3 * t_phase[4]},
            "roofline": {"kernel": "k_interior_fused", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
                         "bytes_per_launch": bytes_launch, "launch_ms": 1e3 * t_launch, "traffic_measured_on": traffic_other},
        }
        # the setup side of the path (Compute seconds and the FP64 matrix-core rate of K6 / K10): flops of one
        # numeric Compute counted from the symbolic plans, over the wall time of the recompute (pattern reused: the
        # numeric work alone); peak = dense FP64 MFMA of MI355X.  MFMA-busy counters of the setup kernels: profiles/.
        out["setup_roofline"] = {"bound": "mfma", "flops": flops_all[0], "flops_factorisations": flops_all[1],
                                 "flops_separator_block_inversions": flops_all[2], "flops_transform_and_dropping": flops_all[3],
                                 "recompute_s": t_recomp, "achieved": flops_all[0] / t_recomp / 1e12 / world if t_recomp > 0 else None,
                                 "peak": 78.6, "unit": "TFLOP/s", "frac": flops_all[0] / t_recomp / 1e12 / world / 78.6 if t_recomp > 0 else None,
                                 "note": "whole numeric Compute (host part included), per GPU; flops from the symbolic plans"}
        if krylov:
            out["krylov"] = krylov
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample: a 48^3 grid has no third level with sx = cx = 8 (and the serial coarse LU of a larger
            # two-level sample takes minutes to set up, as it does for the reference): two-level sample
            out["cpu_baseline"] = cpu_baseline(args.problem, args.re, args.cpu_n, sx, min(levels, 1), gpu_lib_ok=not args.hostsim)
        else:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
