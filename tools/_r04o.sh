mkdir -p gpurun_out/r04o; O=gpurun_out/r04o
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fullsize.py tests/test_block_map.py -m gpu -q --tb=short -x 2>&1 | grep -v "^\s*$" | cut -c1-300 | tail -12) > $O/tests.log; tail -3 $O/tests.log
for w in C2 C3 C4; do for i in 1 2; do
timeout 200 python bench.py --no-train --no-cpu-baseline --steps 20 --workload $w >> $O/$w.json 2>> $O/err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04o/*.json")):
    for l in open(f):
        if l.startswith("{"):
            j=json.loads(l); print(f.split("/")[-1], "%.3f ms/step"%j["ms_per_step"], "bwd %.3f fwd %.3f" % (j["kernel_ms"]["march_backward"], j["kernel_ms"]["march_forward"]), "render", round(j["render"]["ms"],3))
PY
