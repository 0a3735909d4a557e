"""Per-dispatch durations of the merged level solves (k_lvl_fwd / k_lvl_bwd) of the LAST ApplyInverse in a rocprofv3 kernel trace.
usage: python3 tools/trace_apply_levels.py <run_kernel_trace.csv> <launches per ApplyInverse of each kernel>"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void hymls::dev::", "").replace("hymls::dev::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
# the last ApplyInverse: from the last k_interior_fused pair backwards
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_interior_fused")]
first = idx[-2]
seg = rows[first:]
t0 = seg[0][0]
print("last ApplyInverse: %d dispatches, span %.3f ms" % (len(seg), (seg[-1][1] - t0) / 1e6))
prev_end = None
for s, e, n, g in seg:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +%7.1f us  gap %6.1f us  %-28s workgroups %d" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:28], g))
    prev_end = e
