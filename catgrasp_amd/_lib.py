"""ctypes binding of libcatgrasp_amd.so (the C ABI declared in include/catgrasp_amd.h).

The HIP library is the product: there is no CPU fallback.  `lib()` raises if the shared object
is missing or does not export every declared symbol; every wrapper raises on a non-zero status.
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CATGRASP_AMD_LIB', os.path.join(_PKG, 'libcatgrasp_amd.so'))   # override: dev ablation builds only
HEADER_PATH = os.path.join(_PKG, '..', 'include', 'catgrasp_amd.h')
_lib = None


class CatgraspAmdError(RuntimeError):
    pass


def declared_symbols():
    """Names of every function declared in include/catgrasp_amd.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cg_[a-z0-9_]+)\s*\(', src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CatgraspAmdError(
                f'{LIB_PATH} not found: build it with `python -m catgrasp_amd.build` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        l = ctypes.CDLL(LIB_PATH)
        missing = [s for s in declared_symbols() if not hasattr(l, s)]
        if missing:
            raise CatgraspAmdError(f'libcatgrasp_amd.so lacks symbols {missing}; rebuild it')
        l.cg_version.restype = ctypes.c_char_p
        _lib = l
    return _lib


def _p(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object through four
    layers of python (8 us under a profiler, per launch -- a tenth of a 16-candidate predict_batch call); the raw handle is what
    torch's own compiled backends fetch."""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:          # a torch build without the private accessor: the public (slower) route
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status, what):
    if status != 0:
        raise CatgraspAmdError(f'{what} failed with status {status}')


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CatgraspAmdError('catgrasp_amd kernels need CUDA/HIP device tensors (no CPU fallback)')


def f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def i32c(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return t
