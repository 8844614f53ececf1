#!/bin/bash
# round-6 GPU call 23: ray generation with v_rcp / v_rsq instead of twelve IEEE divisions, camera through scalar loads: the whole GPU
# suite (ray tolerances 1e-6 / 2e-6 / 2e-5; the fused camera entry point must stay bit-identical), A/B of the step and of `render`.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06u; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-workloads"
for i in 1 2 3; do
  timeout 200 python tools/bench_variant.py build_variants/libmvp_r06y.so $M 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', d['ms_per_step'], d['kernel_ms'], 'render', d['render']['ms'], 'fused', d['fused_rays_step']['ms'])" | tee -a $O/ab.txt
  timeout 200 python bench.py $M 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', d['ms_per_step'], d['kernel_ms'], 'render', d['render']['ms'], 'fused', d['fused_rays_step']['ms'])" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pn; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o t -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-train --no-render > /dev/null 2>&1 < /dev/null
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pn/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:]:
    if 'raydirs' in r[0] or 'march_kernel<false' in r[0]: print(r[0][:60], r[1], r[3], r[5])
PY
