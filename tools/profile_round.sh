#!/bin/bash
# Runs on the GPU box (via gpurun): (1) the default bench.py line, counter CSVs of its live rocprofv3 --pmc passes kept;
# (2) the same command under rocprofv3 --kernel-trace --stats (no counters in that run), condensed by
# tools/summarize_trace.py.  usage: tools/profile_round.sh <tag> [bench.py arguments]     outputs under gpurun_out/<tag>_*
set -o pipefail
TAG=${1:-r03}
shift || true
EXTRA="$*"   # further bench.py arguments (--workload chr1 ...)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py $EXTRA --keep-pmc "$OUT/${TAG}_pmc" > "$OUT/${TAG}_bench_full.json" 2> "$OUT/${TAG}_bench_full.err" || { echo "bench failed"; tail -20 "$OUT/${TAG}_bench_full.err"; exit 1; }
tail -8 "$OUT/${TAG}_bench_full.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" $EXTRA --no-pmc --cpu-seconds 0 > "$OUT/${TAG}_bench_trace.json" 2> "$OUT/${TAG}_bench_trace.err" || { echo "trace run failed"; tail -20 "$OUT/${TAG}_bench_trace.err"; exit 1; }
cd "$ROOT"
python3 tools/summarize_trace.py "$OUT/${TAG}_trace" "$OUT/${TAG}_bench_trace.json" > "$OUT/${TAG}_rocprof_summary.json" && head -c 1500 "$OUT/${TAG}_rocprof_summary.json"
