#!/bin/bash
# round-6 GPU call 17: one bound per channel group (colour / alpha) in the backward's fixed-point scales: the whole GPU suite, the
# two-pass marks of a training backward with the background MLP on, A/B of the march legs against the previous build, train legs.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06r; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
timeout 300 python tools/diag_two_pass_marks.py C3 bg 2>&1 < /dev/null | tail -5 | tee $O/diag_C3_bg.txt
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in "C2" "C2 --alpha-gain 40" "C3" "C4"; do
    MVP_VARIANT_ABI=17 timeout 200 python tools/bench_variant.py build_variants/libmvp_r06prune.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
for i in 1 2; do
  timeout 300 python bench.py --mode train --workload C3 --steps 16 --warmup 5 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('C3', d['value'], {k: round(v,3) for k,v in d['train']['kernel_ms'].items() if 'march' in k})" | tee -a $O/train.txt
done
timeout 400 python bench.py --mode train --workload C2 --bg on --steps 10 --warmup 3 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('C2_bg', d['value'], {k: round(v,3) for k,v in d['train']['kernel_ms'].items() if 'bgmlp' in k or 'march' in k})" | tee -a $O/train.txt
