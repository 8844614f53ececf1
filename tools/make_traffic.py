#!/usr/bin/env python3
"""tools/make_traffic.py <fetch_summary.csv> <write_summary.csv> [workload] -- derive profiles/traffic.json (per-launch HBM
bytes of the march kernels, read by bench.py for `roofline.traffic`) from the two separate rocprofv3 --pmc passes that
tools/pmc.sh summarised, and STAMP it with the commit the passes were measured at (run this in the build container
right after the gpurun call returned, before the kernels change again).  bench.py prints the stamp next to the number.

Bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(MI355X_MICROARCH.md, "HBM"); both counters are in units of 1024 B."""
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
    doc["_measured_at_commit"] = head + ("+uncommitted kernel changes" if dirty else "")
    doc["_sources"] = [os.path.relpath(os.path.abspath(p), ROOT) for p in (fetch, write)]
    doc[workload] = {k: (2.0 * f[k] + w[k]) * 1024.0 for k in sorted(f) if k in w}
    json.dump(doc, open(tf, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
