"""Condenses rocprofv3 CSV output (kernel-trace stats + PMC passes) into a small text/JSON summary that is
committed under profiles/.  usage: summarize_profile.py <dir made by tools/profile_bench.sh>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


# kernel stats
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    keep = [r for i, r in enumerate(rows) if i < 8 or any(t in r["Name"] for t in ("count_nt2", "locate", "count_scalar"))]
    out["kernel_stats"] = [{k: (r[k][:160] if k == "Name" else r[k]) for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in keep]

# per-dispatch durations of the hot kernel from the kernel trace
for f in find("trace", "*kernel_trace.csv"):
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ranked = sorted(d.items(), key=lambda kv: -sum(kv[1]))
    keep = [kv for i, kv in enumerate(ranked) if i < 8 or any(t in kv[0] for t in ("count_nt2", "locate", "count_scalar"))]
    out["kernel_trace_avg_us"] = {k[:90]: {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
                                  for k, v in keep}

# PMC passes: counter value per dispatch, averaged per kernel
for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
    for f in find(sub, "*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        res = {}
        for k, cs in agg.items():
            if "count_nt2" in k or "locate" in k or "count_scalar" in k:
                res[k[:90]] = {c: {"dispatches": len(v), "avg": sum(v) / len(v)} for c, v in cs.items()}
        out[sub] = res

print(json.dumps(out, indent=1))
