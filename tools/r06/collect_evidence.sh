#!/bin/bash
# tools/r06/collect_evidence.sh <tag> -- in the build container, after `gpurun -- 'bash tools/evidence.sh <tag>'` returned: copy the
# summaries into profiles/ and derive traffic.json (per workload), the binding table and the MFMA fractions from them.
set -eu
TAG=$1; cd "$(dirname "$0")/../.."
G=gpurun_out
cp $G/$TAG/bench.json profiles/${TAG}_bench.json
cp $G/$TAG/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv 2>/dev/null || cp $G/$TAG/*kernel_stats.csv profiles/ 2>/dev/null || true
for t in fetch write sq ta lds tcc lanes fetch_C3 write_C3 fetch_C4 write_C4 fetch_sat write_sat mfma mfma_c2bg; do
  [ -f $G/${TAG}_$t/pmc_summary.csv ] && cp $G/${TAG}_$t/pmc_summary.csv profiles/${TAG}_pmc_$t.csv
done
cp $G/$TAG/train_C3_kernel_stats.csv profiles/${TAG}_train_C3_kernel_stats.csv 2>/dev/null || true
cp $G/$TAG/bgmlp_bench.json profiles/${TAG}_bgmlp_bench.json 2>/dev/null || true
cp $G/$TAG/warp_bench.json profiles/${TAG}_warp_bench.json 2>/dev/null || true
python tools/make_traffic.py profiles/${TAG}_pmc_fetch.csv profiles/${TAG}_pmc_write.csv C2 profiles/${TAG}_pmc_sq.csv profiles/${TAG}_pmc_lds.csv > /dev/null
python tools/make_traffic.py profiles/${TAG}_pmc_fetch_C3.csv profiles/${TAG}_pmc_write_C3.csv C3 > /dev/null
python tools/make_traffic.py profiles/${TAG}_pmc_fetch_C4.csv profiles/${TAG}_pmc_write_C4.csv C4 > /dev/null
python tools/make_traffic.py profiles/${TAG}_pmc_fetch_sat.csv profiles/${TAG}_pmc_write_sat.csv C2_saturated > /dev/null
python tools/make_mfma.py profiles/${TAG}_pmc_mfma.csv > /dev/null
python tools/make_mfma.py profiles/${TAG}_pmc_mfma_c2bg.csv C2_bg > /dev/null
python -c "import json; d=json.load(open('profiles/traffic.json')); print({k: d[k] for k in ('C2','C3','C4','C2_saturated')}); print(d['valu']); print(d['mfma'])"
