#!/bin/bash
# tools/pmc_cmd.sh <tag> "<counters>" <kernel-name-substring> -- <command...>: per-dispatch averages of PMC counters for matching kernels
set -u
TAG=$1; CTRS=$2; PAT=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o $TAG -- "$@" > gpurun_out/$TAG/cmd.log 2>&1
CSV=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$CSV" "$PAT" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:50]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg:
    print(k, "dispatches", len(cnt[k]), {c: "%.4g" % (v / len(cnt[k])) for c, v in sorted(agg[k].items())})
PY
