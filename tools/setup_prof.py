"""Per-shape timing of the big-front launches of the numeric setup (HYMLS_MI_SETUP_PROF=1 in device_hip.hip).
usage: python3 tools/setup_prof.py N SX LEVELS [PROBLEM]
Runs tools/pmc_driver.py (Initialize + one Compute, no ApplyInverse) in a child process with the profiler on (every scope
is synchronised, so the run is slower than a normal Compute) and prints totals per launch kind and the heaviest shapes."""
import os, subprocess, sys
from collections import defaultdict
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:4] + ["0"] + sys.argv[4:5]
env = dict(os.environ, HYMLS_MI_SETUP_PROF="1")
r = subprocess.run([sys.executable, os.path.join(here, "pmc_driver.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
print(r.stdout.strip())
if r.returncode:
    print(r.stderr[-4000:])
    sys.exit(r.returncode)
rows = []
for line in r.stderr.splitlines():
    if not line.startswith("SETUPPROF "):
        continue
    t = line.split()
    rows.append((t[1], tuple(int(v) for v in t[2:7]), int(t[8]), float(t[10]), float(t[12])))
tot = defaultdict(lambda: [0, 0.0, 0.0])
for name, key, calls, ms, tf in rows:
    k = name if name != "gemm" else "gemm<%03d>%s" % (key[0], " big" if key[1] > 96 and key[2] > 96 else "")
    tot[k][0] += calls; tot[k][1] += ms; tot[k][2] += tf * ms * 1e9
print("%-22s %8s %10s %9s" % ("launch", "calls", "ms", "TFLOP/s"))
for k, (c, ms, fl) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %8d %10.1f %9.2f" % (k, c, ms, fl / (ms * 1e9) if ms else 0))
print("\nheaviest shapes (kind, shape..., batch): calls, ms, TFLOP/s")
for name, key, calls, ms, tf in sorted(rows, key=lambda r: -r[3])[:40]:
    print("%-12s %-32s %6d %9.2f %8.2f" % (name, key, calls, ms, tf))
# the Schur-complement update by size class
print("\ngemm<100> (trailing update) by M range: calls, ms, TFLOP/s")
cls = defaultdict(lambda: [0, 0.0, 0.0])
for name, key, calls, ms, tf in rows:
    if name == "gemm" and key[0] == 100:
        b = 0 if key[1] <= 96 else 1 if key[1] <= 256 else 2 if key[1] <= 512 else 3 if key[1] <= 1024 else 4
        cls[b][0] += calls; cls[b][1] += ms; cls[b][2] += tf * ms * 1e9
for b in sorted(cls):
    c, ms, fl = cls[b]
    print("  M %-10s %7d %9.1f %8.2f" % (["<=96", "97-256", "257-512", "513-1024", ">1024"][b], c, ms, fl / (ms * 1e9)))
