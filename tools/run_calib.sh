#!/bin/bash
# on the GPU box: gather calibration, plain and under rocprofv3 --pmc FETCH_SIZE (separate pass, kernel-trace only)
# build first (works without a GPU): mkdir -p tools/bin && hipcc --offload-arch=gfx950 -O3 tools/calib_gather.hip -o tools/bin/calib_gather
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/calib; mkdir -p $OUT
for MB in 128 2048 16384; do ./tools/bin/calib_gather $MB 256 | tee $OUT/calib_${MB}.txt; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_2048 -- ./tools/bin/calib_gather 2048 256 > $OUT/pmc_2048.txt 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc2_2048 -- ./tools/bin/calib_gather 2048 256 > $OUT/pmc2_2048.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/calib/pmc_2048", "gpurun_out/calib/pmc2_2048"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
