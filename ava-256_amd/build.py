"""In-tree hipcc build of libmvp_gfx950.so (gfx950 only; hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmvp_gfx950.so")
SOURCES = ["raydirs.hip", "aabb.hip", "march_fwd.hip", "march_bwd.hip", "march_host.hip", "assemble.hip", "placement.hip", "gradclip.hip", "bgmlp.hip", "pixeltail.hip", "primpose.hip", "abi_misc.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("mvp_device.h", "mvp_host.h", "march_common.h", "march_packet.h")] + [
    os.path.join(ROOT, "include", "mvp_abi.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain missing)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library next to this file. Returns its path."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


VARIANT_DIR = os.path.join(ROOT, "build_variants")


def build_variant(name, defines, force=False, extra_flags=()):
    """Compile a NON-product variant of the library into build_variants/libmvp_<name>.so (tests and timing
    experiments only, e.g. defines=["MVP_DEBUG_HOOKS"] for the build that honours the MVP_DEBUG_* environment)."""
    os.makedirs(VARIANT_DIR, exist_ok=True)
    out = os.path.join(VARIANT_DIR, "libmvp_%s.so" % name)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + ["-D" + d for d in defines] + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out + ".tmp"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
