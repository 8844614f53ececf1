cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05g; mkdir -p $O
# workgroups per CU of bwd_prim_kernel through the debug build's LDS pad (22.5 KB per 2-wave workgroup at C2; 160 KB per CU)
for R in 1 2; do for PAD in 0 4608 10240 18432 32768; do
  MVP_DEBUG_LDS_PAD=$PAD timeout 300 python tools/bench_variant.py build_variants/libmvp_dbg.so --steps 10 --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lds pad $PAD round $R: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/occupancy.txt
done; done
