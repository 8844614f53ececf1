#!/usr/bin/env python3
"""tools/make_mfma.py <pmc_mfma_summary.csv> [workload] -- record the MFMA utilisation of the train leg's dense kernels in
profiles/traffic.json ("mfma": per kernel SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), the fraction of the
SIMD-cycles of the kernel in which its MFMA pipe was busy, against the NOMINAL peak as the contract asks).  The summary comes
from tools/pmc_all.sh on `bench.py --mode train --workload C3` (tools/evidence.sh); bench.py prints the record as
`train.<leg>.mfma_frac` -- a recorded measurement with the commit stamp of the evidence run, like `roofline.traffic`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "C3"
    ctr = {}
    for line in list(open(path))[1:]:
        k, _, c, _, per = line.rstrip("\n").rsplit(",", 4)
        ctr.setdefault(k, {})[c] = float(per)
    out = {}
    for k, v in ctr.items():
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            name = ("bgmlp_fwd_kernel" if "bgmlp::fwd" in k else "bgmlp_bwd_kernel" if "bgmlp::bwd" in k else
                    "hipblaslt_" + k.split("UserArgs_")[-1][:24].rstrip("_"))
            frac = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            if frac >= 0.02:
                out[name] = round(frac, 4)
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(tf)) if os.path.exists(tf) else {}
    out["_source"] = os.path.relpath(os.path.abspath(path), ROOT)  # (per workload: the two legs come from two passes)
    doc.setdefault("mfma", {})[workload] = out
    doc["_mfma_source"] = out["_source"]
    json.dump(doc, open(tf, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
