"""Kernel time of ONE numeric recompute from a rocprofv3 kernel trace of `tools/pmc_driver.py` run with PMC_DRIVER_COMPUTES=3.
usage: python3 tools/trace_recompute.py <run_kernel_trace.csv>
The second and third Compute launch the same kernel sequence; the trace is cut at the last occurrence of that period, so
the table covers exactly the last Compute: per kernel calls / ms, the GPU-busy total and the wall span (first start to last
end: the difference is time the device waited for the host)."""
import csv, re, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# period: the longest suffix that repeats immediately before itself
n = len(names)
best = 0
for L in range(n // 2, 50, -1):
    if names[n - L:] == names[n - 2 * L:n - L]:
        best = L
        break
if not best:
    print("no repeating suffix found; kernels", n)
    sys.exit(1)
last = rows[n - best:]
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
print("last Compute: %d dispatches, GPU busy %.3f s, span %.3f s (device idle %.3f s)" % (best, busy / 1e9, span / 1e9, (span - busy) / 1e9))
# idle gaps by size
gaps = [last[i + 1][0] - last[i][1] for i in range(len(last) - 1)]
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps %8.0f-%-12.0f ns: %7d, %.3f s" % (lo, hi, len(g), sum(g) / 1e9))
tot = defaultdict(lambda: [0, 0])
for s, e, k in last:
    k = re.sub(r"\(.*", "", k).replace("void hymls::dev::", "").replace("hymls::dev::", "")
    tot[k][0] += 1; tot[k][1] += e - s
print("%-44s %8s %10s %6s" % ("kernel", "calls", "ms", "%"))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-44s %8d %10.2f %6.1f" % (k[:44], c, t / 1e6, 100.0 * t / busy))
