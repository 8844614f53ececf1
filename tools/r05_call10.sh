cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05l; mkdir -p $O
for Q in 1 2 4; do python tools/exp_compact_check.py build_variants/libmvp_cmpq$Q.so 2>&1 | tail -1 | tee -a $O/check.txt; done
for R in 1 2; do
  for WL in C2 C3 C4; do
    for V in gfx950 cmpq1:0 cmpq1:1 cmpq2:0 cmpq2:1 cmpq4:1; do
      N=${V%%:*}; C=${V##*:}; L=build_variants/libmvp_$N.so; [ $N = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so && C=0
      MVP_COMPACT=$C timeout 300 python tools/bench_variant.py $L --steps 10 --workload $WL --no-workloads 2>$O/err_${N}_$C.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$N compact=$C $WL round $R: step %.3f fwd %.3f bwd %.3f render %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward'], d['render']['ms']))" | tee -a $O/ab.txt
    done
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in gfx950:0 cmpq2:1; do
  N=${V%%:*}; C=${V##*:}; L=build_variants/libmvp_$N.so; [ $N = gfx950 ] && L=ava-256_amd/libmvp_gfx950.so
  for CT in "FETCH_SIZE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
    MVP_COMPACT=$C bash tools/pmc_cmd.sh r05k_$N "$CT" march_ -- python tools/bench_variant.py $L --steps 3 --warmup 1 --no-render 2>&1 | grep "march_kernel<false\|march_cull" | tee -a $O/counters.txt
  done
done
