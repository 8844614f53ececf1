cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05u; mkdir -p $O
timeout 60 python tools/exp_sched_check.py build_variants/libmvp_sched.so 2>&1 | tail -1 | tee $O/check.txt
grep -q "True" $O/check.txt || exit 0
for R in 1 2; do for WL in C2 C3 C4; do for C in 0 1; do
  MVP_SCHED=$C timeout 60 python tools/bench_variant.py build_variants/libmvp_sched.so --steps 10 --workload $WL --no-workloads 2>$O/err_$C.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('sched=$C $WL round $R: step %.3f fwd %.3f bwd %.3f render %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward'], d['render']['ms']))" | tee -a $O/ab.txt
done; done; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in 0 1; do for CT in "FETCH_SIZE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  MVP_SCHED=$C timeout 90 bash tools/pmc_cmd.sh r05u_$C "$CT" march_kernel -- python tools/bench_variant.py build_variants/libmvp_sched.so --steps 3 --warmup 1 --no-render 2>&1 | grep "march_kernel<false" | sed "s/^/sched=$C /" | tee -a $O/counters.txt
done; done
