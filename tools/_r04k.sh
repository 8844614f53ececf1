mkdir -p gpurun_out/r04k; O=gpurun_out/r04k
(timeout 600 python -m pytest tests/test_trainloop.py tests/test_trainstep_parity.py tests/test_assemble.py -m gpu -q --tb=short 2>&1 | grep -v "^\s*$" | cut -c1-300 | tail -30) > $O/tests.log; tail -3 $O/tests.log
for i in 1 2; do timeout 300 python bench.py --mode train --workload C3 --steps 12 --warmup 3 >> $O/train_C3.json 2>> $O/err.log; done
timeout 300 python bench.py --mode train --workload C2 --steps 6 --warmup 2 >> $O/train_C2.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k/train_*.json")):
    for l in open(f):
        if l.startswith("{"):
            j=json.loads(l); print(f.split("/")[-1], "%.2f it/s  %.3f ms/iter" % (j["value"], j["ms_per_step"]), {k: round(v,3) for k,v in j["train"]["kernel_ms"].items()})
PY
timeout 300 python tools/profile_train_ops.py C3 1 > $O/ops_C3.txt 2>&1; grep -v "^\[W\|Warn\|warn" $O/ops_C3.txt | head -50 | cut -c1-50,100-180
bash tools/fwd_strip_l2.sh r04k > $O/strip.log 2>&1; tail -7 $O/strip.log
