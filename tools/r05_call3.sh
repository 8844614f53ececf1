#!/bin/bash
# round-5 GPU call 3: full GPU suite on the product (packed 64-bit scatter, fp16 render path), the timing decomposition of
# the new backward (no atomics / raw-bit atomics / no march), counters of the fp16 render kernel.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_product.log 2>&1; tail -4 $O/pytest_product.log
cp gpurun_out/parity_masks.json $O/parity_masks.json 2>/dev/null
for R in 1 2; do for V in gfx950 exp2 exp7 exp3; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  timeout 300 python tools/bench_variant.py $L --steps 10 --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V C2 round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/decomp.txt
done; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/half_render.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench, argparse
a = argparse.Namespace(workload="C2", alpha_gain=1.0, cams=None, scaling="weak")
step, info = bench.make_march_step_gpu(a, 0, 1, torch.device("cuda", 0))
with torch.no_grad():
    for _ in range(4): info["render_half"]()
    for _ in range(4): info["render"]()
torch.cuda.synchronize()
PY
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -c1-12 | tr ' ' '_')
  bash tools/pmc_cmd.sh r05c_half_$T "$C" march -- python /tmp/half_render.py 2>&1 | tee -a $O/half_counters.txt
done
