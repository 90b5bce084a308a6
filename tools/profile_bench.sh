#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench's hot kernel.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-seconds 0 --no-variants $*"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log" || { echo "trace run failed"; tail -20 "$OUT/bench_trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.log" || { echo "fetch pmc failed"; tail -20 "$OUT/bench_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/bench_write.log" || { echo "write pmc failed"; tail -20 "$OUT/bench_write.log"; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_l2" -- python3 bench.py $ARGS > "$OUT/bench_l2.json" 2> "$OUT/bench_l2.log" || { echo "l2 pmc failed"; tail -20 "$OUT/bench_l2.log"; }
find "$OUT" -name "*.csv" | head -40
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
