"""far_amd.ops._base: streams, pointers, workspaces and the activation-range state shared by every front end (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags


_TLS = threading.local()          # .side: the library side stream this thread's launches currently go to (ops.side), or absent

def _stream():
    s = getattr(_TLS, 'side', None)
    return s if s is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

class side:
    """`with ops.side(i):` -- the launches inside go to the library's side stream i, which first waits for everything the
    current torch stream holds (far_stream_fork); they overlap with what the caller launches afterwards until ops.join(i).
    torch's allocator knows only the current stream, so every tensor the side launches touch must outlive the join: locals of
    the caller do; temporaries of the ops called inside are appended to a `keep` list by those ops (their `keep=` argument)."""

    def __init__(self, i):
        self.i = i

    def __enter__(self):
        h = _lib.load().far_stream_fork(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), self.i)
        if not h:
            raise _lib.FarHipError('far_stream_fork failed')
        self.prev = getattr(_TLS, 'side', None)
        _TLS.side = ctypes.c_void_p(h)

    def __exit__(self, *exc):
        _TLS.side = self.prev
        return False

def join(i):
    """The current torch stream waits for side stream i (far_stream_join)."""
    _lib.check(_lib.load().far_stream_join(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), i), 'far_stream_join')

def _p(t, dtype=None):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    if not t.is_contiguous():
        raise _lib.FarHipError('far_amd ops need contiguous tensors')
    if dtype is not None and t.dtype != dtype:
        raise _lib.FarHipError(f'expected dtype {dtype}, got {t.dtype}')
    return ctypes.c_void_p(t.data_ptr())

def _written(t):
    """Tell autograd's version counter that `t` was overwritten through its raw device pointer (the kernels write
    caller-provided `out=` tensors behind torch's back; anything keyed on Tensor._version -- PackCache, the head's
    feature reuse -- must see it).  Inference tensors carry no version counter."""
    if t is not None and not t.is_inference():
        torch.autograd.graph.increment_version(t)
    return t

def tensor_version(t):
    """Tensor._version, or None for inference tensors (created under torch.inference_mode(): immutable outside it,
    no counter to read)."""
    return None if t.is_inference() else t._version

def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)

def _layout(t):
    """0 = contiguous NCHW, 1 = channels_last memory; anything else is re-laid out as NCHW by the caller."""
    if t.is_contiguous():
        return 0
    if t.is_contiguous(memory_format=torch.channels_last):
        return 1
    return -1

def _same_layout(ts):
    """Bring 4-D fp32 GPU tensors to one memory layout (the first tensor's, NCHW if it has neither)."""
    for t in ts:
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
            raise _lib.FarHipError('backbone epilogues need fp32 GPU (N, C, H, W) tensors')
    lay = _layout(ts[0])
    if lay < 0:
        lay = 0
    fmt = torch.channels_last if lay == 1 else torch.contiguous_format
    return lay, [t if _layout(t) == lay else t.contiguous(memory_format=fmt) for t in ts]

def _raw(t):
    """Device pointer of a tensor whose memory layout the caller has established (channels_last 4-D tensors are not `contiguous`)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())

_ACT = {'none': 0, 'relu': 1, 'leaky': 2}

_CONV_ACT_EXP = 4          # default exponent of the activation scale 2^e applied before the fp16 split (conv_igemm_f16s.hip);

                           # the packed `scale` vectors fold 2^-4, the kernel corrects for the exponent actually used


# ---------------------------------------------------------------------------------------------------------------------
# Activation range of the split-precision kernels (K9, K13, K14).  An fp32 activation a is split hi = fp16(a 2^e),
# lo = fp16(a 2^e - hi): |a| <= 65504 / 2^e survives, beyond it hi = inf.  The reference's fp32 convolutions / Linear layers
# have no such limit (resnet_fpn.py:101-119, transformer.py:44-67), so the limit must never bite silently:
#   * every launch tests its accumulators and ORs a per-device flag (no cost unless it fires);
#   * `check_activation_range` reads the flag (one host read) and raises ActivationOverflow;
#   * far_amd.loftr.LoFTR catches it, lowers the exponent e (thread-local, `activation_exponent`) by 4 -- 16x the range,
#     16x coarser absolute resolution of values below 2^-14 2^-e -- switches the fused fine-level layers (whose exponent
#     is fixed) to their K9 + K5 form, and re-runs the forward.  e = 4 covers |a| <= 4094; the floor e = -24 covers 1e12.
# ---------------------------------------------------------------------------------------------------------------------
class ActivationOverflow(_lib.FarHipError):
    pass

_ACT_STATE = threading.local()

_OVERFLOW_FLAGS = {}

ACT_EXP_MIN = -24

def activation_exponent_value():
    return getattr(_ACT_STATE, 'exp', _CONV_ACT_EXP)

class activation_exponent:
    """Context manager: K9 launches of this thread split their activations around 2^exp."""

    def __init__(self, exp):
        if not (ACT_EXP_MIN <= int(exp) <= 8):
            raise ValueError(f'activation exponent must be in [{ACT_EXP_MIN}, 8]')
        self.exp = int(exp)

    def __enter__(self):
        self.prev = activation_exponent_value()
        _ACT_STATE.exp = self.exp
        return self

    def __exit__(self, *a):
        _ACT_STATE.exp = self.prev

def overflow_flag(device):
    """The per-device int32 flag every K9 / K13 / K14 launch ORs into."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    f = _OVERFLOW_FLAGS.get(key)
    if f is None:
        f = _OVERFLOW_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=torch.device('cuda', key))
    return f

def activation_overflowed(device, reset=True):
    """True if a launch since the last reset saw a non-finite accumulator.  One blocking host read."""
    f = overflow_flag(device)
    hit = bool(f.item())
    if hit and reset:
        f.zero_()
    return hit

def check_activation_range(device, what='far_amd'):
    if activation_overflowed(device):
        raise ActivationOverflow(
            f'{what}: an activation left the range of the split-fp16 operands (|a| > {65504.0 / 2.0 ** activation_exponent_value():.4g} '
            f'at activation exponent {activation_exponent_value()}): the outputs of this call contain inf / NaN.  '
            f'Re-run under ops.activation_exponent(e) with a lower e (far_amd.loftr.LoFTR does this by itself).')

def grad_scale(x):
    """Two device floats { 2^e, 2^(4 - e) }, max|x| 2^e in [2^9, 2^10): the activation scale of a K9 launch whose input is a
    gradient (conv_nhwc(..., act_scale_dev=...)); no host synchronisation."""
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(lib.far_grad_scale_f32(_p(x, torch.float32), x.numel(), _p(out), _stream()), 'far_grad_scale_f32')
    return out
