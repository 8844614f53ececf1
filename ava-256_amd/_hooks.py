"""Optional instrumentation hooks of the operators (tests / bench only; both default to off)."""
import torch

from . import _lib

diag = None    # device int32[8] the march kernels accumulate MVP_DIAG_* counters into (include/mvp_abi.h)
events = None  # list collecting (name, start_event, end_event) per C-ABI launch
keep_raysat = False                 # tests: keep the last forward's raysat tensor in `last_raysat`
last_raysat = None
last_pl_count = None                # ... and its forward->backward hand-off counters ([N*K] counts, then flags, bounds)
last_flags_index = 0                # index of the flags word in it (N*K)
last_handoff_shape = None           # (N, H, W, K, list capacity) of that forward


class patched_handoff:
    """Tests / tools: run the forwards inside the `with` block with another hand-off allocation -- `cap=...`: that list
    capacity per primitive (a small one pushes most primitives to the ray-centric kernel); `ray_centric=True`: no hand-off
    buffers at all, so the ray-centric kernel owns everything.  Works by replacing mvpraymarch.alloc_handoff: the operator
    itself reads no test switch."""

    def __init__(self, cap=None, ray_centric=False):
        self.cap, self.ray_centric = cap, ray_centric

    def __enter__(self):
        import importlib
        m = importlib.import_module(__package__ + ".mvpraymarch")  # (the package re-exports a FUNCTION of that name)
        self._m, self._orig = m, m.alloc_handoff
        cap, ray_centric = self.cap, self.ray_centric

        def alloc(N, H, W, K, dev):
            if ray_centric:
                return None, None, None, 0
            rayaux, pl_count, _, _ = self._orig(N, H, W, K, dev)
            return rayaux, pl_count, torch.empty((N * K, cap, m.LIST_ENTRY_WORDS), device=dev, dtype=torch.int32), cap

        if cap is not None or ray_centric:
            m.alloc_handoff = alloc
        return self

    def __exit__(self, *exc):
        self._m.alloc_handoff = self._orig
        return False


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
