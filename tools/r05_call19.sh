cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05v; mkdir -p $O
for V in sched4x4 sched4x2 sched2x2 sched8x2; do for WL in C2 C3; do
  MVP_SCHED=1 timeout 60 python tools/bench_variant.py build_variants/libmvp_$V.so --steps 10 --workload $WL --no-render 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V $WL: step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward']))" | tee -a $O/ab.txt
done; done
