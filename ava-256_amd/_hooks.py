"""Optional instrumentation hooks of the operators (tests / bench only; both default to off)."""
import torch

from . import _lib

diag = None    # device int32[8] the march kernels accumulate MVP_DIAG_* counters into (include/mvp_abi.h)
events = None  # list collecting (name, start_event, end_event) per C-ABI launch
force_ray_centric_backward = False  # tests: skip the forward->backward hand-off so the fallback kernel runs
primlist_cap_override = None        # tests: force a (small) per-primitive list capacity
keep_raysat = False                 # tests: keep the last forward's raysat tensor in `last_raysat`
last_raysat = None
last_pl_count = None                # ... and its forward->backward hand-off counters ([N*K] counts, then flags, bounds)


def set_diag_buffer(t):
    """Give the march kernels a zeroed int32[8] device tensor to accumulate diagnostics into (or None)."""
    global diag
    if t is not None:
        assert t.is_cuda and t.dtype == torch.int32 and t.numel() >= _lib.DIAG_WORDS and t.is_contiguous()
    diag = t


def read_diag():
    if diag is None:
        return None
    return dict(zip(_lib.DIAG_NAMES, diag.cpu().tolist()))


def set_event_sink(lst):
    """bench.py: collect HIP events around each C-ABI launch (recorded on the stream the kernel runs on)."""
    global events
    events = lst


class timed:
    def __init__(self, name, dev):
        self.name, self.dev = name, dev

    def __enter__(self):
        if events is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))

    def __exit__(self, *exc):
        if events is not None:
            self.b.record(torch.cuda.current_stream(self.dev))
            events.append((self.name, self.a, self.b))
        return False
