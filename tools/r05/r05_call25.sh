#!/bin/bash
# bg MLP: bias through LDS + next tile's pixel requested ahead: parity tests, fused time, per-dispatch kernel durations
O=gpurun_out/r05x4; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -3 $O/tests.log
timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2>$O/bench.err < /dev/null; cat $O/bgmlp_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_bgmlp_fused.py 4 512 512 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "bgmlp" in n or n.startswith("Cijk"):
        d[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v2 = sorted(v)
    print(k, len(v), "us min %.0f med %.0f max %.0f" % (v2[0], v2[len(v2) // 2], v2[-1]), " all:", " ".join("%.0f" % x for x in v[:16]))
PY
