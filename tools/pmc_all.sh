#!/bin/bash
# tools/pmc_all.sh <tag> "<counters>" [bench args...] -- like tools/pmc.sh but keeps EVERY kernel of the run (the train leg's
# GEMMs are hipBLASLt/rocBLAS kernels, not mvp:: ones): per-kernel sums of each counter -> gpurun_out/<tag>/pmc_summary.csv
set -u
TAG=$1; shift
CTRS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o $TAG -- python bench.py "$@" > gpurun_out/$TAG/bench.log 2>&1
CSV=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$CSV" gpurun_out/$TAG/pmc_summary.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
order = sorted(agg, key=lambda k: -max(agg[k].values()))
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches,counter,sum,per_dispatch\n")
    for k in order[:40]:
        for c, v in sorted(agg[k].items()):
            line = "%s,%d,%s,%.6g,%.6g" % (k.replace(",", ";"), len(cnt[k]), c, v, v / max(1, len(cnt[k])))
            f.write(line + "\n")
for k in order[:12]:
    print(k[:60], {c: "%.3g" % v for c, v in sorted(agg[k].items())})
PY
