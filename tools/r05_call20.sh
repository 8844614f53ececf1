cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05w; mkdir -p $O
for R in 1 2; do for V in gfx950 pad1 pad3 pad7 pad9 rot1115 rot33; do
  L=build_variants/libmvp_$V.so; [ $V = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  for WL in C2 C4; do
  timeout 60 python tools/bench_variant.py $L --steps 10 --workload $WL --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V $WL round $R: bwd %.3f' % (d['kernel_ms']['march_backward']))" | tee -a $O/ab.txt
  done
done; done
