#!/bin/bash
# kernel-trace stats of the whole N=1 bench incl. the locate pipeline and the variants (run on the GPU box)
set -o pipefail
TAG=${1:-r01}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_locate_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 > "$OUT/bench.json" 2> "$OUT/bench.log" || { tail -20 "$OUT/bench.log"; exit 1; }
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt"; python3 - <<PY
import json
d=json.load(open("$OUT/summary.txt"))
for k,v in d["kernel_trace_avg_us"].items():
    if any(t in k for t in ("count_nt2","locate","count_scalar")): print(k[:80], v)
PY
