#!/bin/bash
# kernel-trace + PMC passes of the locate pipeline alone (tools/time_locate.py); run on the GPU box
set -o pipefail
TAG=${1:-r01}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_walk_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/time_locate.py > "$OUT/loc.json" 2> "$OUT/loc.log" || { tail -20 "$OUT/loc.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/time_locate.py > "$OUT/loc2.json" 2> "$OUT/loc2.log" || { tail -20 "$OUT/loc2.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 tools/time_locate.py > "$OUT/loc3.json" 2> "$OUT/loc3.log" || { tail -20 "$OUT/loc3.log"; }
python3 - <<PY
import csv, glob, collections
for sub in ("trace", "pmc_fetch", "pmc_sq"):
    print("==", sub)
    for f in glob.glob("$OUT/%s/**/*kernel_stats.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "locate" in r["Name"] or "count_nt2_reads" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "locate" in r["Kernel_Name"]:
                a = agg[(r["Kernel_Name"][:50], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in sorted(agg.items()): print(k, "launches", n, "avg", v / n)
PY
