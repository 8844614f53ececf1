#!/bin/bash
# round-6 GPU call 35: three waves per primitive (15 list entries per round) forced at C2 -- the TRAINING scene has 13 % of its
# primitives on more than 10 packets (bench scene: 3 %; tools/diag_train_scene.py), i.e. a second round with PW = 2 -- bench scene
# and train leg, product (PW = 2 at C2) against the forced build.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render --no-workloads"
T='import sys, json
from ava256_amd import _lib
if sys.argv[1] != "product": _lib.use_library(sys.argv[1])
import bench
bench.main(["--mode", "train", "--workload", "C2", "--steps", "12", "--warmup", "4", "--bg", "off"])'
for i in 1 2; do
  timeout 200 python bench.py $M --workload C2 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pw2 bench', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/pw3.txt
  timeout 200 python tools/bench_variant.py build_variants/libmvp_pw3.so $M --workload C2 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pw3 bench', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/pw3.txt
  timeout 300 python -c "$T" product 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); t=d.get('train', d); t=t.get('C2', t); print('pw2 train', t.get('iters_per_s'), t.get('kernel_ms'))" | tee -a $O/pw3.txt
  timeout 300 python -c "$T" build_variants/libmvp_pw3.so 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); t=d.get('train', d); t=t.get('C2', t); print('pw3 train', t.get('iters_per_s'), t.get('kernel_ms'))" | tee -a $O/pw3.txt
done
