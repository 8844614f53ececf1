#!/bin/bash
# round-6 GPU call 2: mask-compacted phase 1 of the backward (16-byte list entries carrying the forward's ray mask):
# smoke + GPU suite, then A/B against round 5's library at C2 / C3 / C4, three interleaved rounds.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06b; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in C2 C3 C4; do
    MVP_VARIANT_ABI=15 timeout 200 python tools/bench_variant.py build_variants/libmvp_r05base.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
