cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05f; mkdir -p $O
for R in 1 2; do for V in gfx950 exp3 fe1 fe3 fe5 fe7 fe15; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  timeout 300 python tools/bench_variant.py $L --steps 10 --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V C2 round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/frontend.txt
done; done
