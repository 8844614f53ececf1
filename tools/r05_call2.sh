#!/bin/bash
# round-5 GPU call 2: full GPU suite on the product (incl. the fp16-slab path), bench line with render_fp16 + secondary legs,
# and two backward candidates (software-pipelined walk; packed 64-bit atomics): parity tests against each, then A/B timing.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_product.log 2>&1; tail -3 $O/pytest_product.log
timeout 600 python bench.py --no-train > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05b/bench.json").read().strip().split("\n")[-1])
print("step", d["ms_per_step"], d["kernel_ms"]); print("render", d["render"]["ms"], "render_fp16", d["render_fp16"]["ms"], d["render_fp16"]["template_to_half_ms"])
print("saturated", d["saturated"]["ms_per_step"], d["saturated"]["kernel_ms"], d["saturated"]["saturated_ray_fraction"])
for k, v in d["workloads"].items(): print(k, v["ms_per_step"], v["kernel_ms"])
PY
for V in pipeA u64; do
  timeout 900 python tools/pytest_variant.py build_variants/libmvp_$V.so tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_hardening.py -m gpu -x -q > $O/pytest_$V.log 2>&1; echo "$V: $(tail -1 $O/pytest_$V.log)"
done
for R in 1 2; do for V in gfx950 pipeA u64; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  for WL in C2 C3 C4; do
    timeout 300 python tools/bench_variant.py $L --steps 10 --no-render --workload $WL 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V $WL round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/ab.txt
  done
done; done
