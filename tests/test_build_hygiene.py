"""Host tests of the product build's hygiene: no timing variant lives in the product sources or
can be switched on through build.FLAGS, the parked variants still apply as a patch, and the product build's per-kernel
registers / LDS / scratch are the recorded ones (profiles/kernel_resources.csv) -- a change of code generation has to be made
on purpose (`python tools/kernel_resources.py --write`)."""
import importlib.util
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ava-256_amd", "csrc")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_build_flags_define_no_knob():
    build = _load(os.path.join(ROOT, "ava-256_amd", "build.py"), "_mvp_build_hygiene")
    assert not [f for f in build.FLAGS if f.startswith("-D")], build.FLAGS
    assert set(build.SOURCES) == {f for f in os.listdir(CSRC) if f.endswith(".hip")}


# knobs that selected timing variants (wrong results by construction) or measured-negative schedules through round 4
PARKED = ["MVP_EXP", "MVP_FWD_QUAD", "MVP_FWD_OCC", "MVP_REC_SLOTS", "MVP_FAST_CROSS", "MVP_FAST_RECS", "MVP_STRIP_ROWS",
          "MVP_NO_STRIP_ORDER", "MVP_NO_STEP_ROTATION", "MVP_ROT_MUL", "MVP_NO_STREAM_HINTS", "MVP_GRADPAD",
          "MVP_ENTRIES_PER_WAVE", "MVP_BWD_OCC", "MVP_PRECISE_WAVES", "BG_EXP"]


def test_product_sources_carry_no_timing_variant():
    for f in sorted(os.listdir(CSRC)):
        text = open(os.path.join(CSRC, f)).read()
        for knob in PARKED:
            assert not re.search(r"\b%s\b" % knob, text), "%s mentions %s" % (f, knob)
        # the only conditional compilation left: include guards / pragma once and the test build's debug hooks
        for m in re.finditer(r"^\s*#\s*(if|ifdef|ifndef|elif)\b(.*)$", text, re.M):
            assert "MVP_DEBUG_HOOKS" in m.group(2), "%s: %s" % (f, m.group(0).strip())


def test_parked_variants_still_apply():
    patch = os.path.join(ROOT, "profiles", "r05_timing_variants.patch")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(CSRC, os.path.join(tmp, "ava-256_amd", "csrc"))
        r = subprocess.run(["patch", "-p1", "-s", "--dry-run", "-i", patch], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_product_build_matches_recorded_kernel_resources():
    from ava256_amd import _lib
    from ava256_amd import build as pkg_build
    pkg_build.build()                       # (no-op when the library is current)
    kr = _load(os.path.join(ROOT, "tools", "kernel_resources.py"), "_mvp_kernel_resources")
    got = kr.table(_lib.LIB_PATH)
    want = open(os.path.join(ROOT, "profiles", "kernel_resources.csv")).read().split("\n")
    want = [l for l in want if l]
    assert got == want, "\n".join(sorted(set(got) ^ set(want)))
    for line in got[1:]:                    # nothing spills to scratch, no kernel spills VGPRs
        f = line.rsplit(",", 7)
        assert int(f[2]) == 0 and int(f[6]) == 0, line
