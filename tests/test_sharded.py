"""-m "not gpu": the sharded path on CPU: world_size 2/4/8 `gloo` ranks, each driving the
TEST-ONLY host simulator through the C ABI with hymls_amd.dist.TorchComm as the transport.  The
assembled result of the sharded ApplyInverse must equal the one-rank result on the same problem
(same partition, same ownership, same summation order => agreement to rounding), which in turn is
checked against the oracle in test_hostsim_parity.py.  The GPU version of the same check (ranks sharing
one card) is tests/test_gpu_parity.py::test_sharded_matches_single_gpu."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # world, eq, nx, ny, nz, sx, levels, cx, partitioner
    (2, "Laplace", 16, 8, 8, 4, 1, -1, "Cartesian"),
    (2, "Stokes-C", 16, 8, 8, 4, 1, -1, "Skew Cartesian"),
    (2, "Stokes-C", 16, 8, 8, 4, 0, -1, "Skew Cartesian"),
    (4, "Laplace", 16, 16, 8, 4, 2, 2, "Cartesian"),
    (8, "Stokes-C", 16, 16, 16, 4, 2, 2, "Skew Cartesian"),
]


def run_worker(world, args, mode, port, timeout=900, env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    cmd += [str(a) for a in args] + [mode]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("DIST_RESULT ")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1][len("DIST_RESULT "):])


@pytest.mark.parametrize("world,eq,nx,ny,nz,sx,levels,cx,part", CASES)
def test_sharded_matches_one_rank(hostsim_lib, world, eq, nx, ny, nz, sx, levels, cx, part):
    res = run_worker(world, (eq, nx, ny, nz, sx, levels, cx, part), "hostsim", 29520 + world)
    assert res["cover_ok"]                       # every row owned by exactly one rank
    assert res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-12
    assert res["repeat_diff"] == 0.0 and res["recompute_diff"] < 1e-12
    assert res["matvec_err"] < 1e-13                                        # sharded K x with imported columns
    assert abs(res["krylov_its_sharded"] - res["krylov_its_one_rank"]) <= 1 and res["krylov_residual"] < 1e-6


@pytest.mark.parametrize("world,eq,nx,ny,nz,sx,levels,cx,part", [
    (2, "Stokes-C", 32, 16, 16, 4, 1, -1, "Skew Cartesian"),
    (8, "Laplace", 32, 32, 32, 4, 2, 2, "Cartesian"),
])
def test_sharded_geometric_halo_prefilter(hostsim_lib, world, eq, nx, ny, nz, sx, levels, cx, part):
    """large grids only look at the subdomains near the rank's box when they build the halo; force that code
    path on a small grid (HYMLS_MI_HALO_ALL_BELOW=0) -- a halo subdomain that was missed would change ownership
    or multiplicities and with them the result."""
    res = run_worker(world, (eq, nx, ny, nz, sx, levels, cx, part), "hostsim", 29560 + world,
                     env_extra={"HYMLS_MI_HALO_ALL_BELOW": "0"})
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-12


def test_sharded_host_exchange_in_rounds(hostsim_lib):
    """setup exchanges larger than the per-round limit are cut into rounds (here: 1 kB rounds)"""
    res = run_worker(4, ("Laplace", 16, 16, 8, 4, 1, -1, "Cartesian"), "hostsim", 29575,
                     env_extra={"HYMLS_MI_HOST_CHUNK_BYTES": "1024"})
    assert res["cover_ok"] and res["rel_err"] < 1e-12 and res["matvec_err"] < 1e-13


@pytest.mark.parametrize("world,eq,nx,ny,nz,sx,levels,cx,part", [
    (2, "Stokes-C", 16, 8, 8, 4, 0, -1, "Skew Cartesian"),
    (4, "Stokes-C", 16, 16, 8, 4, 1, -1, "Skew Cartesian"),
    (8, "Laplace", 16, 16, 16, 4, 2, 2, "Cartesian"),
])
def test_sharded_bordered_matches_one_rank(hostsim_lib, world, eq, nx, ny, nz, sx, levels, cx, part):
    """[K V; W' C] on a sharded handle: every rank passes its rows of V and W; the border scalars are all-reduced, the
    border of the Schur system travels with the halo / hand-off exchanges.  Same (X, S) as on one rank."""
    res = run_worker(world, (eq, nx, ny, nz, sx, levels, cx, part), "hostsim", 29580 + world, env_extra={"HYMLS_TEST_BORDER": "1"})
    assert res["cover_ok"] and res["rel_err"] < 1e-12 and res["border_err"] < 1e-10


@pytest.mark.parametrize("world,args", [
    (3, ("Stokes-C", 48, 16, 16, 8, 1, -1, "Skew Cartesian")),
    (6, ("Laplace", 24, 16, 8, 4, 1, -1, "Cartesian")),
])
def test_rank_counts_that_are_no_power_of_two(world, args):
    """rank_grid spreads the prime factors of the world size over x, y, z (3 -> 3x1x1, 6 -> 3x2x1): the sharded ApplyInverse
    equals the one-rank one (the reference's CreatePIDMap accepts any processor count too, BasePartitioner.cpp:361-586)."""
    from hymls_amd.dist import rank_grid
    assert rank_grid(3) == (3, 1, 1) and rank_grid(6) == (3, 2, 1) and rank_grid(8) == (2, 2, 2) and rank_grid(16) == (4, 2, 2)
    res = run_worker(world, args, "hostsim", 29660 + world, timeout=800)
    assert res["cover_ok"] and res["levels"] == res["levels_sharded"]
    assert res["rel_err"] < 1e-10
