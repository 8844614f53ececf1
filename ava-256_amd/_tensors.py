"""Argument checks shared by the operators (host logic; mirrors the reference's CHECK_INPUT / asserts)."""
import torch


def require_device_f32(name, t, allow_none=False):
    """The reference checks is_cuda + is_contiguous (mvpraymarch.cpp:102-104) and silently reinterprets any
    dtype as float (mvpraymarch.cpp:250-272); here dtype is checked too."""
    if t is None:
        if allow_none:
            return None
        raise RuntimeError("%s must not be None" % name)
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor (there is no CPU path)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def aligned(t):
    """The C ABI wants 16-byte aligned arrays; a contiguous view with an odd storage offset is copied."""
    if t is not None and (t.data_ptr() & 15):
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream
