"""Summarise the two rocprofv3 --pmc passes of tools/pmc_driver.py (FETCH_SIZE, WRITE_SIZE; separate passes, kernel filter
k_interior_fused|k_axpby) into profiles/<tag>_pmc_interior_fused.json + a CSV of the per-dispatch counter values.
gfx950: FETCH_SIZE counts a 128-B line request at 64 B (guide: MI355X_MICROARCH.md, HBM section), so the raw value is
doubled; the factor is checked in the same pass on the streaming kernel k_axpby (16 B read + 8 B written per entry).
usage: python3 tools/pmc_summary.py DIR_FETCH DIR_WRITE DRIVER_LOG TAG"""
import csv, glob, json, os, re, sqlite3, sys

def rows(d):
    db = sqlite3.connect(glob.glob(os.path.join(d, "*.db"))[0])
    return list(db.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection order by start"))

dfetch, dwrite, log, tag = sys.argv[1:5]
line = [l for l in open(log) if l.startswith("PMCDRIVER n ")][-1]
m = re.search(r"PMCDRIVER n (\d+) levels (\[.*\]) bytes_interior_per_launch (\d+) stored (\d+) sparse_equivalent (\d+) n1 (\d+)", line)
n, levels, alg, stored, sparse, n1 = int(m.group(1)), eval(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), int(m.group(6))
out = {"kernel": "k_interior_fused", "n": n, "sx": 8, "levels": len(levels) - 1, "problem": "Stokes3D"}
allrows = []
for cname, d in (("FETCH_SIZE", dfetch), ("WRITE_SIZE", dwrite)):
    R = rows(d)
    fused = [r for r in R if "k_interior_fused" in r[0]]
    # calibration: the level-0 k_axpby (x1 -= A11 \ y1 over the n1 interior unknowns): its grid is the capped one
    ax = [r for r in R if "k_axpby" in r[0] and r[1] >= 8192 * 256]
    raw = sum(r[3] for r in fused) / len(fused) * 1024.0
    raw_ax = sum(r[3] for r in ax) / len(ax) * 1024.0
    expect_ax = (16.0 if cname == "FETCH_SIZE" else 8.0) * n1
    out[cname] = {"dispatches": len(fused), "raw_bytes_per_launch": raw, "avg_duration_ms": sum(r[4] for r in fused) / len(fused) / 1e6,
                  "calibration_k_axpby": {"dispatches": len(ax), "raw_bytes": raw_ax, "true_bytes": expect_ax, "factor": expect_ax / raw_ax}}
    allrows += [(cname,) + r for r in fused + ax]
fetch = 2.0 * out["FETCH_SIZE"]["raw_bytes_per_launch"]
write = out["WRITE_SIZE"]["raw_bytes_per_launch"]
out.update({"hbm_read_bytes": fetch, "hbm_write_bytes": write, "hbm_bytes_per_launch": fetch + write,
            "algorithmic_bytes_per_launch": alg, "stored_factor_bytes_per_launch": stored,
            "sparse_equivalent_factor_bytes_per_launch": sparse, "traffic_over_algorithmic": (fetch + write) / alg,
            "correction": "FETCH_SIZE x 2 (gfx950 counts 128-B requests at 64 B; k_axpby in the same pass: factor %.4f), WRITE_SIZE x 1 (factor %.4f)"
                          % (out["FETCH_SIZE"]["calibration_k_axpby"]["factor"], out["WRITE_SIZE"]["calibration_k_axpby"]["factor"]),
            "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex 'k_interior_fused|k_axpby' --kernel-trace -- "
                       "python3 tools/pmc_driver.py %d 8 %d 3 (separate passes; driver is torch-free)" % (n, len(levels) - 1)})
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_interior_fused.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", tag + "_pmc_counter_collection.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["counter", "kernel", "grid_size", "counter_name", "value_KB", "duration_ns"]); w.writerows(allrows)
print(json.dumps(out, indent=1))
