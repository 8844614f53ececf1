#!/usr/bin/env python3
"""tools/make_traffic.py <fetch_summary.csv> <write_summary.csv> [workload [sq_summary.csv lds_summary.csv]] -- workload = C2
(default), C3, C4 or C2_saturated (the same C2 step at opacity x 40: bench.py's `saturated` leg).  Derive profiles/traffic.json (per-launch HBM
bytes of the march kernels, read by bench.py for `roofline.traffic`) from the two separate rocprofv3 --pmc passes that
tools/pmc.sh summarised, and STAMP it with the commit the passes were measured at (run this in the build container
right after the gpurun call returned, before the kernels change again).  bench.py prints the stamp next to the number.

Bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(MI355X_MICROARCH.md, "HBM"); both counters are in units of 1024 B.

With the SQ pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_ANY) and the pass that holds GRBM_GUI_ACTIVE
(the kernel's duration in shader-clock cycles), a "valu" object is recorded as well -- what actually binds these kernels:
  busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)   (the counter ticks in quad-cycles, per SIMD, summed)
  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES                              (share of a wave's life spent in s_waitcnt)
  wave_insts = SQ_INSTS_VALU per launch (wave-level VALU instructions)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, counter):
    out = {}
    for line in list(open(path))[1:]:  # kernel,dispatches,counter,sum,per_dispatch -- kernel names contain commas
        k, _, ctr, _, per = line.rstrip("\n").rsplit(",", 4)
        if ctr != counter:
            continue
        name = "march_forward" if "march_kernel<false" in k else "march_backward" if "bwd_prim_kernel" in k else None
        if name:
            out[name] = out.get(name, 0.0) + float(per)
    return out


def main():
    fetch, write = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "C2"
    f, w = per_dispatch(fetch, "FETCH_SIZE"), per_dispatch(write, "WRITE_SIZE")
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "ava-256_amd/csrc"], capture_output=True,
                           text=True).stdout.strip() != ""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(tf)) if os.path.exists(tf) else {}
    doc["_how"] = ("per-launch HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate --pmc passes (per-dispatch "
                   "averages of the march kernels; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM'); tools/make_traffic.py")
    stamp = head + ("+uncommitted kernel changes" if dirty else "")
    srcs = [os.path.relpath(os.path.abspath(p), ROOT) for p in (fetch, write)]
    if workload == "C2":   # the headline workload keeps the top-level stamp the bench line prints
        doc["_measured_at_commit"] = stamp
        doc["_sources"] = list(srcs)
    # (round 6) every workload carries its own stamp and sources: C3 / C4 / "C2_saturated" are separate passes
    doc.setdefault("_by_workload", {})[workload] = {"measured_at_commit": stamp, "sources": srcs}
    doc[workload] = {k: (2.0 * f[k] + w[k]) * 1024.0 for k in sorted(f) if k in w}
    if len(sys.argv) > 5:
        sq, lds = sys.argv[4], sys.argv[5]
        ctr = lambda path, name: per_dispatch(path, name)
        insts, active, wcyc, wait = (ctr(sq, c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"))
        cyc = {k: v / 8.0 for k, v in ctr(lds, "GRBM_GUI_ACTIVE").items()}  # (the counter is summed over the 8 XCDs)
        doc.setdefault("valu", {})[workload] = {
            k: {"wave_insts": insts[k], "busy": active[k] * 4.0 / (1024.0 * cyc[k]), "wait": wait[k] / wcyc[k],
                "kernel_cycles": cyc[k]} for k in sorted(insts) if k in active and k in cyc and k in wcyc and k in wait}
        more = [os.path.relpath(os.path.abspath(p), ROOT) for p in (sq, lds)]
        doc["_by_workload"][workload]["sources"] += more
        if workload == "C2":
            doc["_sources"] += more
    json.dump(doc, open(tf, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
