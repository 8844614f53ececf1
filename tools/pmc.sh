#!/bin/bash
# tools/pmc.sh <tag> "<counters>" [bench args...] -- one rocprofv3 PMC pass (own run, kernel-trace only) of bench.py;
# prints per-kernel sums of each counter and keeps the aggregated CSV under gpurun_out/<tag>/.
set -u
TAG=$1; shift
CTRS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out/$TAG
timeout 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o $TAG -- python bench.py "$@" > gpurun_out/$TAG/bench.log 2>&1
CSV=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$CSV" gpurun_out/$TAG/pmc_summary.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches,counter,sum,per_dispatch\n")
    for k in agg:
        if "mvp" not in k: continue   # torch's own kernels are not evidence for anything here
        for c, v in agg[k].items():
            line = "%s,%d,%s,%.6g,%.6g" % (k, len(cnt[k]), c, v, v / max(1, len(cnt[k])))
            f.write(line + "\n")
            if "mvp" in k: print(line)
PY
