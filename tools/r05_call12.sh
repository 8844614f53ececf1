cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05n; mkdir -p $O
for R in 1 2; do
  for WL in C2 C3 C4; do
    for V in gfx950:0 cmpq1:0 cmpq1:1 cmpq2:0 cmpq2:1 cmpq4:0 cmpq4:1; do
      N=${V%%:*}; C=${V##*:}; L=build_variants/libmvp_$N.so; [ $N = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
      MVP_COMPACT=$C timeout 300 python tools/bench_variant.py $L --steps 10 --workload $WL --no-render 2>$O/err_${N}_$C.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$N compact=$C $WL round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/ab.txt
    done
  done
done
