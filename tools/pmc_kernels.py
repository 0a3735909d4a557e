"""Per-kernel summary of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run: dispatches, mean duration and
the mean of every collected counter, one line per kernel name (template arguments kept).  Markdown on stdout.
usage: python3 tools/pmc_kernels.py <dir with *_counter_collection.csv> [title]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else d
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
assert files, "no counter_collection.csv under " + d
stat = defaultdict(lambda: {"n": set(), "dur": 0.0, "cnt": defaultdict(float)})
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
        s = stat[name]
        key = (r["Dispatch_Id"], f)
        if key not in s["n"]:
            s["n"].add(key)
            s["dur"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        s["cnt"][r["Counter_Name"]] += float(r["Counter_Value"])
counters = sorted({c for s in stat.values() for c in s["cnt"]})
print("# " + title)
print("| kernel | dispatches | mean duration (us) | " + " | ".join("mean " + c for c in counters) + " |")
print("|---|---|---|" + "---|" * len(counters))
for name, s in sorted(stat.items(), key=lambda kv: -kv[1]["dur"]):
    n = len(s["n"])
    print("| `%s` | %d | %.1f | " % (name, n, s["dur"] / n / 1e3) + " | ".join("%.4g" % (s["cnt"][c] / n) for c in counters) + " |")
