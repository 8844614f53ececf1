#!/bin/bash
# round-6 GPU call 18: where do the +2...5 % of the per-channel-group bounds come from?  Kernel stats of the C2 and C4 march legs
# (new build vs previous) and the marks the backward leaves.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M="--steps 8 --warmup 3 --no-cpu-baseline --no-train --no-render"
for wl in C2 C4; do
  rm -rf /tmp/pn; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o t -- python bench.py $M --workload $wl > /dev/null 2>&1 < /dev/null
  echo "new $wl"; python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pn/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:]:
    if 'mvp' in r[0]: print(r[0][:70], r[1], r[3], r[5])
PY
  rm -rf /tmp/po; MVP_VARIANT_ABI=17 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o t -- python tools/bench_variant.py build_variants/libmvp_r06prune.so $M --workload $wl > /dev/null 2>&1 < /dev/null
  echo "prev $wl"; python - <<'PY'
import csv, glob
f = glob.glob('/tmp/po/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:]:
    if 'mvp' in r[0]: print(r[0][:70], r[1], r[3], r[5])
PY
done
python - <<'PY'
import sys, argparse, torch
sys.path.insert(0, '.')
import bench
for wl in ("C2", "C4"):
    a = argparse.Namespace(workload=wl, cams=None, scaling="weak", alpha_gain=1.0, gout="randn")
    step, info = bench.make_march_step_gpu(a, 0, 1, torch.device("cuda", 0))
    print(wl, info["marks"]())
    del step, info; torch.cuda.empty_cache()
PY
