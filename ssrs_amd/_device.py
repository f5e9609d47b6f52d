"""Device plumbing: torch owns HBM buffers and streams, nothing else."""
import numpy as np
import torch


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('ssrs_amd needs a ROCm GPU (MI355X/gfx950): '
                           'torch.cuda.is_available() is False and there is no CPU fallback')


def device():
    require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    """hipStream_t of torch's current stream, so torch.cuda.Event sees our kernels."""
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def to_dev(x, dtype=None):
    """numpy / tensor -> contiguous CUDA tensor (optionally cast)."""
    if x is None:
        return None
    if is_tensor(x):
        t = x if x.is_cuda else x.to(device())
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device())
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def float_dev(x):
    """Keep f32/f64 rasters in their own precision; everything else -> f64."""
    t = to_dev(x)
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    return t


def like_input(t, ref):
    """Return `t` as numpy when the caller passed numpy, else the tensor."""
    return t if is_tensor(ref) else t.cpu().numpy()


def ftype(t):
    from . import _native
    return _native.SSRS_F64 if t.dtype == torch.float64 else _native.SSRS_F32
