#!/bin/bash
# tools/fwd_stage_census.sh <tag> -- per-stage instruction census of the forward march at C2 (on the GPU box, via gpurun): the
# debug build (build_variants/libmvp_dbg.so, -DMVP_DEBUG_HOOKS) stops every packet after a stage (MVP_DEBUG_STAGE: 11 root test,
# 12 ancestor pre-cull, 13 implicit level, 1 traversal, 2 exact test + crossing tables, 3 sweep without sampling, 0 everything);
# one rocprofv3 --pmc pass per stage.  Output: gpurun_out/<tag>/fwd_stage_insts.csv (cumulative per launch).
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
echo "stage,meaning,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_INSTS_SALU,SQ_WAVE_CYCLES,SQ_WAIT_ANY,GRBM_GUI_ACTIVE" > $O/fwd_stage_insts.csv
for st in 11 12 13 1 2 3 0; do
  case $st in 11) m="rays + packet bounds + root test + stores";; 12) m="+ ancestor pre-cull";; 13) m="+ implicit depth-10 level";; 1) m="+ compacted levels (traversal done)";; 2) m="+ exact test, crossing tables, list hand-off";; 3) m="+ sweep without sampling";; 0) m="everything";; esac
  OUT=/tmp/pmc_stage_$st; rm -rf $OUT
  MVP_DEBUG_STAGE=$st timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o s -- python tools/bench_variant.py build_variants/libmvp_dbg.so --steps 2 --warmup 1 --no-render > $O/stage_$st.log 2>&1
  python - "$(find $OUT -name '*counter_collection.csv' | head -1)" "$st" "$m" >> $O/fwd_stage_insts.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    if "march_kernel<false" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n = max(1, len(disp))
print(",".join([sys.argv[2], sys.argv[3]] + ["%.5g" % (agg[c] / n) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")]))
PY
done
cat $O/fwd_stage_insts.csv
