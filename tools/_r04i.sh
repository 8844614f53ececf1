mkdir -p gpurun_out/r04i; O=gpurun_out/r04i
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -m gpu -q --tb=short -x 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
for i in 1 2; do timeout 200 python bench.py --no-train --no-cpu-baseline --no-render --steps 20 >> $O/C2.json 2>> $O/err.log; done
for w in C3 C4; do timeout 200 python bench.py --no-train --no-cpu-baseline --no-render --steps 20 --workload $w >> $O/$w.json 2>> $O/err.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04i/*.json")):
    for l in open(f):
        if l.startswith("{"):
            j=json.loads(l); print(f.split("/")[-1], "%.3f ms/step"%j["ms_per_step"], "bwd %.3f fwd %.3f" % (j["kernel_ms"]["march_backward"], j["kernel_ms"]["march_forward"]))
PY
bash tools/pmc_cmd.sh r04i_bwd "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" bwd_prim_kernel -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render
bash tools/fwd_stage_census.sh r04i
