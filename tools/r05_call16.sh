cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05s; mkdir -p $O
for R in 1 2; do for V in gfx950 occ4 occ4w8 epw3; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  for WL in C2 C3 C4; do
  timeout 300 python tools/bench_variant.py $L --steps 10 --workload $WL --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V $WL round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/ab.txt
  done
done; done
timeout 900 python tools/pytest_variant.py build_variants/libmvp_occ4w8.so tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_hardening.py -m gpu -x -q > $O/pytest_occ4w8.log 2>&1; echo "occ4w8: $(tail -1 $O/pytest_occ4w8.log)"
