"""-m "not gpu": the N > 1 path of bench.py (one process per rank, barrier + max-over-ranks
timing, aggregate value on rank 0) with world_size 2 on the CPU: gloo backend + the TEST-ONLY
host simulator.  The numbers are meaningless (and marked so); the plumbing is what is tested."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--grid", "16", "--sx", "4", "--levels", "1", "--hostsim", "--no-cpu-baseline"]
    cmd += list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    return json.loads(lines[0])


def test_bench_two_ranks_sharded_gloo():
    r = _run(2)
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["unit"] == "DoF/s" and r["higher_is_better"] is True and r["scaling"] == "strong" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and "TEST ONLY" in r["data"]
    assert "sharded: 2x1x1 boxes of 8x16x16" in r["config"]["parallelism"] and "16x16x16" in r["config"]["workload"]
    assert r["config"]["levels"][0][1] == 16 * 16 * 16 * 4      # the one global problem, split over the ranks
    assert r["value"] > 0 and r["ms_per_step"] > 0
    assert set(r["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert r["cpu_baseline"] is None   # rank 0 at N=1 only


def test_bench_two_ranks_replicas_gloo():
    r = _run(2, ["--replicas"])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert "2 replicas" in r["config"]["parallelism"] and "workload" in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0
    assert set(r["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert r["cpu_baseline"] is None   # rank 0 at N=1 only


def test_bench_single_rank_fields():
    r = _run(1)
    assert r["n_gpus"] == 1 and "1 GPU" in r["config"]["parallelism"]
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["peak"] == 8000.0
