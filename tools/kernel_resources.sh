#!/bin/bash
# tools/kernel_resources.sh -- (from SOURCE; tools/kernel_resources.py reads the same table out of the built library in
# half a second and is what tests/test_build_hygiene.py and /tmp/kernel_resources_from_source.csv use) per-kernel register / LDS / scratch usage of every HIP source (compiled for gfx950 exactly
# as build.py does: keep the code-generation flags below equal to build.py's FLAGS), from the code-object metadata.  Output: /tmp/kernel_resources_from_source.csv
set -eu
cd "$(dirname "$0")/.."
TMP=$(mktemp -d)
echo "source,kernel,vgpr,vgpr_spill,sgpr,sgpr_spill,lds_bytes,scratch_bytes,waves_per_simd_by_vgpr" > /tmp/kernel_resources_from_source.csv
for f in raydirs aabb march_fwd march_bwd assemble placement gradclip bgmlp pixeltail primpose abi_misc; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -I include -I ava-256_amd/csrc -S --cuda-device-only \
        ava-256_amd/csrc/$f.hip -o $TMP/$f.s 2>/dev/null
  python3 - "$TMP/$f.s" "$f" >> /tmp/kernel_resources_from_source.csv <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in txt else ""
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace(",", ";")
    v = g("vgpr_count")
    waves = 8 if v <= 64 else 7 if v <= 72 else 6 if v <= 80 else 5 if v <= 96 else 4 if v <= 128 else 3 if v <= 168 else 2 if v <= 256 else 1
    print("%s,%s,%d,%d,%d,%d,%d,%d,%d" % (sys.argv[2], dem, v, g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                                        g("group_segment_fixed_size"), g("private_segment_fixed_size"), waves))
PY
done
rm -rf $TMP
cat /tmp/kernel_resources_from_source.csv
