cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05r; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for R in 1 2 3; do for V in base gfx950; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  for WL in C2 C3 C4; do
  timeout 300 python tools/bench_variant.py $L --steps 10 --workload $WL --no-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V $WL round $R: step %.3f fwd %.3f bwd %.3f render %.3f render_fp16 %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward'], d['render']['ms'], d['render_fp16']['ms']))" | tee -a $O/ab.txt
  done
done; done
