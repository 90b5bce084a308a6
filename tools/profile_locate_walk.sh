#!/bin/bash
# kernel-trace stats of the locate pipeline alone (tools/time_locate.py); run on the GPU box.
# (PMC passes over this script spend minutes in the index construction's thousands of small launches: use a smaller
#  text -- tools/time_locate.py takes the text length as its first argument -- when counters are needed.)
set -o pipefail
TAG=${1:-r01}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_walk_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/time_locate.py "$@" > "$OUT/loc.json" 2> "$OUT/loc.log" || { tail -20 "$OUT/loc.log"; exit 1; }
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "locate" in r["Name"] or "localise" in r["Name"] or "count_nt2_reads" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
