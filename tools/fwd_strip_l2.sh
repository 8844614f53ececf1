#!/bin/bash
# tools/fwd_strip_l2.sh <tag> -- forward march at C2 with dispatch strips of 1 / 2 / 3 (product) / 4 / 6 packet rows
# (build_variants/libmvp_strip<r>.so = tools/build_patched.sh strip<r> profiles/r05_timing_variants.patch -DMVP_STRIP_ROWS=r): time, L2 hit rate and HBM read traffic per launch.  Question: does a
# strip height whose in-flight slabs fit the XCD's 4 MB L2 remove the 1.78x over-fetch, and does that matter?
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
echo "strip_rows,fwd_ms,TCC_HIT_sum,TCC_MISS_sum,l2_hit_rate,FETCH_SIZE_KB,hbm_read_GB(2xFETCH)" > $O/fwd_strip_l2.csv
for r in 1 2 3 4 6; do
  if [ $r = 3 ]; then CMD="python bench.py"; else CMD="python tools/bench_variant.py build_variants/libmvp_strip$r.so"; fi
  ms=$(timeout 200 $CMD --steps 20 --no-train --no-cpu-baseline --no-render 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads([l for l in sys.stdin if l.startswith('{')][-1])['kernel_ms']['march_forward'])")
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    OUT=/tmp/pmc_strip; rm -rf $OUT
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o s -- $CMD --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-render > /dev/null 2>&1
    python - "$(find $OUT -name '*counter_collection.csv' | head -1)" >> $O/strip_$r.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    if "march_kernel<false" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
for k, v in agg.items(): print(k, v / max(1, len(disp)))
PY
  done
  python - $r $ms $O/strip_$r.txt >> $O/fwd_strip_l2.csv <<'PY'
import sys
d = dict((l.split()[0], float(l.split()[1])) for l in open(sys.argv[3]))
h, m, f = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0), d.get("FETCH_SIZE", 0)
print("%s,%s,%.4g,%.4g,%.4f,%.4g,%.3f" % (sys.argv[1], sys.argv[2], h, m, h / max(1.0, h + m), f, 2 * f * 1024 / 1e9))
PY
done
cat $O/fwd_strip_l2.csv
