import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle64():
    from oracle.mvp_oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.mvp_oracle import Oracle
    return Oracle("f32")


def pytest_sessionfinish(session, exitstatus):
    """What the parity tests masked, recorded: every use of helpers.FragileRays appends
    {config, rays, hitting_rays, masked_saturation, masked_edge, bound}; the fuzz adds its escape-hatch count.  Written next to
    the other per-call artefacts of a GPU run so that it travels back (gpurun_out/ is merged into the build container)."""
    try:
        import json
        import helpers
        recs = list(helpers.MASK_RECORDS)
        if not recs:
            return
        import test_gpu_parity
        out = {"masked": recs, "fuzz_escapes": list(getattr(test_gpu_parity, "FUZZ_ESCAPES", [])),
               "fuzz_seeds": os.environ.get("MVP_FUZZ_SEEDS"), "hit_frac_bound": helpers.HIT_FRAC,
               "worst_fraction_of_hitting_rays": max((r["masked_saturation"] / r["hitting_rays"]) for r in recs if r["hitting_rays"])
               if any(r["hitting_rays"] for r in recs) else None}
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_masks.json"), "w") as f:
            json.dump(out, f, indent=1)
    except Exception as e:  # a bookkeeping failure must not turn a green run red
        print("parity_masks.json not written:", e)
