#!/bin/bash
# tools/evidence.sh <tag> -- the per-round evidence run on the GPU box (via gpurun): rocprofv3 kernel stats of the default
# bench line, separate FETCH_SIZE / WRITE_SIZE / SQ / TA / LDS counter passes of the march kernels, and kernel stats + MFMA
# counters of the train leg (C3: 4 frames, fused bf16 background MLP), the MLP and warp-field microbenchmarks.  Everything lands in gpurun_out/<tag>*/ ; copy what is to
# be judged into profiles/ and run `python tools/make_traffic.py profiles/<tag>_pmc_fetch.csv profiles/<tag>_pmc_write.csv C2 <sq> <lds>` in the
# build container (it stamps traffic.json with the commit; the SQ and LDS summaries give `roofline.valu`), the same with
# `..._fetch_C3.csv ..._write_C3.csv C3`, `C4` and `..._fetch_sat.csv ..._write_sat.csv C2_saturated` for the other legs (tools/r06/collect_evidence.sh does all of it), and
# `python tools/make_mfma.py profiles/<tag>_pmc_mfma.csv` for `train.C3.mfma_frac` (and `... <tag>_pmc_mfma_c2bg.csv C2_bg` for the 80-frame leg).
set -u
TAG=$1
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
bash tools/prof.sh $TAG --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-render > $O/prof.log 2>&1; tail -6 $O/prof.log
M="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render"
bash tools/pmc.sh ${TAG}_fetch "FETCH_SIZE" $M > $O/fetch.log 2>&1
bash tools/pmc.sh ${TAG}_write "WRITE_SIZE" $M > $O/write.log 2>&1
bash tools/pmc.sh ${TAG}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $M > $O/sq.log 2>&1
bash tools/pmc.sh ${TAG}_ta "TA_TA_BUSY_sum TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" $M > $O/ta.log 2>&1
bash tools/pmc.sh ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $M > $O/lds.log 2>&1
bash tools/pmc.sh ${TAG}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $M > $O/tcc.log 2>&1
bash tools/pmc.sh ${TAG}_lanes "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" $M > $O/lanes.log 2>&1
# (round 6) FETCH / WRITE passes of the other march workloads of the bench line: C3, C4 and the saturated C2 scene
for wl in C3 C4; do
  bash tools/pmc.sh ${TAG}_fetch_$wl "FETCH_SIZE" $M --workload $wl > $O/fetch_$wl.log 2>&1
  bash tools/pmc.sh ${TAG}_write_$wl "WRITE_SIZE" $M --workload $wl > $O/write_$wl.log 2>&1
done
bash tools/pmc.sh ${TAG}_fetch_sat "FETCH_SIZE" $M --alpha-gain 40 > $O/fetch_sat.log 2>&1
bash tools/pmc.sh ${TAG}_write_sat "WRITE_SIZE" $M --alpha-gain 40 > $O/write_sat.log 2>&1
bash tools/pmc_all.sh ${TAG}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" --mode train --workload C3 --steps 4 --warmup 2 > $O/mfma.log 2>&1
bash tools/pmc_all.sh ${TAG}_mfma_c2bg "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" --mode train --workload C2 --bg on --steps 2 --warmup 1 > $O/mfma_c2bg.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_train; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python bench.py --mode train --workload C3 --steps 6 --warmup 2 > $O/train_prof.log 2>&1
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} $O/train_C3_kernel_stats.csv \;
python - $O/train_C3_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:40]]
csv.writer(open(sys.argv[1], "w")).writerows(out)
for r in out[:12]: print(" | ".join(x[:60] for x in r[:5]))
PY
timeout 300 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2>/dev/null; cat $O/bgmlp_bench.json
timeout 300 python tools/bench_warp.py 4 512 512 4096 > $O/warp_bench.json 2>/dev/null; cat $O/warp_bench.json
bash tools/pmc_cmd.sh ${TAG}_bgmlp "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_pmc.log 2>&1; tail -3 $O/bgmlp_pmc.log
cut -c1-200 $O/bench.json
