#!/bin/bash
# round-6 GPU call 7: deep primitives stay in bwd_prim_kernel (rounds cut to their share of the noise budget) instead of going to
# the two-pass kernel: full GPU suite, the two-pass diagnostics and kernel stats of the C2 train legs again, the default bench
# line's new legs (train_like), and the march A/B against the previous build.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06g; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
timeout 300 python tools/diag_two_pass_marks.py C2 2>&1 < /dev/null | tail -6 | tee $O/diag_C2.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for bg in off on; do
  rm -rf /tmp/prof_t; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python bench.py --mode train --workload C2 --bg $bg --steps 6 --warmup 3 > $O/train_C2_$bg.log 2>&1 < /dev/null
  find /tmp/prof_t -name "*kernel_stats.csv" -exec cp {} $O/train_C2_${bg}_kernel_stats.csv \;
  python - $O/train_C2_${bg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:100]] + r[1:] for r in rows[1:30]]
csv.writer(open(sys.argv[1], "w")).writerows(out)
for r in out[:12]:
    if "mvp" in r[0] or r[0] == "Name": print(" | ".join(x[:64] for x in r[:6]))
PY
done
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2; do
  for wl in C2 C3 C4; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06pose.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pose', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc $?"; python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_ms"])
print("train_like", {k: d["train_like"][k] for k in ("ms_per_step", "kernel_ms", "backward_marks")})
print("saturated", d["saturated"]["kernel_ms"], "C3", d["workloads"]["C3"]["kernel_ms"], "C4", d["workloads"]["C4"]["kernel_ms"])
for k, v in d["train"].items():
    if isinstance(v, dict): print(k, v["iters_per_s"], {a: round(b, 3) for a, b in v["kernel_ms"].items() if "march" in a})
print(d["cpu_baseline"]["sample"])
PY
