cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_half.py tests/test_trainloop.py -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for R in 1 2 3; do timeout 300 python bench.py --steps 5 --no-train --no-cpu-baseline --no-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('round $R: render %.3f render_fp16 %.3f to_half %.3f step %.3f' % (d['render']['ms'], d['render_fp16']['ms'], d['render_fp16']['template_to_half_ms'], d['ms_per_step']))" | tee -a $O/half.txt; done
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
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  T=$(echo $C | cut -c1-12 | tr ' ' '_')
  bash tools/pmc_cmd.sh r05q_half_$T "$C" march -- python /tmp/half_render.py 2>&1 | tee -a $O/half_counters.txt
done
