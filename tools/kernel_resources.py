#!/usr/bin/env python3
"""tools/kernel_resources.py [library.so] [--write] -- per-kernel register / LDS / scratch usage of the BUILT library, read
from the code-object metadata inside it (no recompilation): the .hip_fatbin section is a sequence of clang offload bundles
(one per translation unit), each carrying one gfx950 ELF whose AMDGPU note lists every kernel's resources.

Prints the table; with --write also stores it as profiles/kernel_resources.csv.  tests/test_build_hygiene.py compares
the product build with that file, so a change of any kernel's registers, LDS or scratch shows up as a test failure
until the table is regenerated on purpose."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HEADER = "kernel,vgpr,vgpr_spill,sgpr,sgpr_spill,lds_bytes,scratch_bytes,waves_per_simd_by_vgpr"


def waves_by_vgpr(v):
    """512 VGPRs per SIMD lane on gfx950, allocated in blocks of 8: waves = floor(512 / roundup(v, 8)), at most 8."""
    return max(1, min(8, 512 // max(8, (v + 7) // 8 * 8)))


def device_elfs(lib):
    """Every gfx950 code object embedded in `lib`, as bytes."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    return out


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True,
                             check=True).stdout
    if "amdhsa.kernels:" not in txt:
        return []
    rows = []
    for blk in txt[txt.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
        def g(k):
            return int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace(",", ";")
        v = g("vgpr_count")
        rows.append("%s,%d,%d,%d,%d,%d,%d,%d" % (dem, v, g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                                                 g("group_segment_fixed_size"), g("private_segment_fixed_size"),
                                                 waves_by_vgpr(v)))
    return rows


def table(lib):
    rows = []
    for elf in device_elfs(lib):
        rows += kernels_of(elf)
    return [HEADER] + sorted(rows)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--write"]
    lib = args[0] if args else os.path.join(ROOT, "ava-256_amd", "libmvp_gfx950.so")
    t = table(lib)
    print("\n".join(t))
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", "kernel_resources.csv"), "w") as f:
            f.write("\n".join(t) + "\n")
