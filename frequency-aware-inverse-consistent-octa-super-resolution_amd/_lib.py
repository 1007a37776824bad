"""ctypes binding of the C ABI declared in include/faoctasr.h.

The product path has no CPU or eager-PyTorch fallback: if libfaoctasr.so is missing or a
tensor is not a contiguous fp32 device tensor, calls fail loudly."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FAOCTASR_LIB") or os.path.join(_HERE, "libfaoctasr.so")      # override: kernel-variant experiments (tools/variants.py)

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_SIG = {
    # name: (restype, "argument codes")   p pointer, i int, l long, f float
    "version": (_I, ""),
    "last_error": (ctypes.c_char_p, ""),
    "last_route": (_I, ""),
    "conv_wpack_floats": (_L, "iiiiiiii"),
    "conv_pack_job": (_L, "p l i pp iiiiiiiii i i i"),
    "conv_pack_run": (_I, "p i l p"),
    "conv_pack_scales": (_I, "p i p"),
    "absmax_bits": (_I, "p l p p"),
    "conv_set_scales": (_I, "pp"),
    "conv_needs_scales": (_I, "i iiiiiiiii i i"),
    "out_absmax": (_I, "p"),
    "conv_set_workspace": (_I, "p l"),
    "conv_set_residual": (_I, "p"),
    "conv_wgrad_workspace_floats": (_L, "iiiii"),
    "conv2d_fwd": (_I, "pppp iiiiiiiii i i f p i i p"),
    "conv2d_dgrad": (_I, "ppp iiiiiiiii p i i p"),
    "conv2d_wgrad": (_I, "ppp iiiiiiiii i i i p"),
    "conv_transpose2d_fwd": (_I, "pppp iiiiiiiii i i f p i i p"),
    "conv_transpose2d_dgrad": (_I, "ppp iiiiiiiii i p i i p"),
    "conv_transpose2d_wgrad": (_I, "ppp iiiiiiiii i i i p"),
    "reflect_pad_bwd": (_I, "pp iiii p"),
    "channel_sum": (_I, "pp iii i p"),
    "bn_workspace_floats": (_L, "i"),
    "batchnorm_train_fwd": (_I, "ppppppppp iii ff i f p p"),
    "batchnorm_train_bwd": (_I, "ppppppppppp iii i f i p p"),
    "batchnorm_eval_fwd": (_I, "pppppp iii f i f p"),
    "batchnorm_eval_bwd": (_I, "ppppp iii f i f p"),
    "instancenorm_fwd": (_I, "pppppp iii f i f p p"),
    "instancenorm_bwd": (_I, "pppppppppp iii i f p p"),
    "act_fwd": (_I, "pp l i f p"),
    "act_bwd": (_I, "ppp l i f p"),
    "cat2_act_fwd": (_I, "ppp iiii i f p"),
    "cat2_act_bwd": (_I, "pppp iiii i f p"),
    "axpby": (_I, "ppp l ff p"),
    "haar_dwt2d_fwd": (_I, "ppp l ii p"),
    "haar_dwt2d_bwd": (_I, "ppp l ii p"),
    "haar_dfront_fwd": (_I, "pp iii i p"),
    "haar_dfront_bwd": (_I, "pp iii i p"),
    "sgemm_batched": (_I, "ppp iii iii lll i p"),
    "freq_mix_fwd": (_I, "ppppp l p"),
    "freq_mix_bwd": (_I, "pppppppp l p"),
    "ssim_fwd": (_I, "ppp iiii p"),
    "ssim_bwd": (_I, "ppp i f pp iiii p"),
    "loss_workspace_floats": (_L, ""),
    "loss_fwd": (_I, "ppp l i f p p"),
    "loss_bwd": (_I, "pppp l i f i p"),
    "mean_mix_fwd": (_I, "ppp iii ff p"),
    "mean_mix_bwd": (_I, "ppp iii ff p"),
    "adamw_step": (_I, "pppp l fffff i f p"),
    "adamw_step_dev": (_I, "pppp l p p"),
    "prep_crop_resize": (_I, "pppp iiiii ff p"),
    "fill": (_I, "p l f p"),
    "circulant_lowpass": (_I, "p i f p"),
    "eval_workspace_bytes": (_L, "ii"),
    "eval_metrics": (_I, "pppp iii f i p"),
    "bn_fold": (_I, "pppppp f pp i l i l p"),
    "comm_unique_id": (_I, "p"),
    "comm_create": (_I, "p ii p"),
    "comm_destroy": (_I, "p"),
    "comm_size": (_I, "p"),
    "grad_allreduce": (_I, "p l i p p"),
    "param_broadcast": (_I, "p l i p p"),
}
_CODE = {"p": _P, "i": _I, "l": _L, "f": _F}
PACK_JOB_BYTES = 1024          # include/faoctasr.h FAOCTASR_PACK_JOB_BYTES

_lib = None


class KernelError(RuntimeError):
    pass


def declared_symbols():
    return ["faoctasr_" + k for k in _SIG]


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KernelError("libfaoctasr.so is not built (%s); run `python __graft_entry__.py` or "
                          "`python frequency-aware-inverse-consistent-octa-super-resolution_amd/build.py` -- "
                          "there is no CPU/eager fallback for the product path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, codes) in _SIG.items():
        fn = getattr(lib, "faoctasr_" + name)
        fn.restype = res
        fn.argtypes = [_CODE[c] for c in codes.replace(" ", "")]
    _lib = lib
    return lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of the calling thread's current stream on its current device.  ``torch.cuda.current_stream()`` builds a Stream
    object through three Python layers (7.5 us a call, ~800 calls per step and thread: 6 ms of a 36 ms batch-1 step by cProfile,
    tools/host_profile.py); the raw accessors are two C calls."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise KernelError("kernel operand must be a contiguous fp32 device tensor, got %s %s contiguous=%s on %s"
                          % (tuple(t.shape), t.dtype, t.is_contiguous(), t.device))
    return t.data_ptr()


#: optional launch timer (bench.py installs one): an object with ``names`` and ``add(name, args, start_event, end_event)``
launch_timer = None


_fn = {}


def call(name, *args):
    fn = _fn.get(name)
    if fn is None:
        fn = _fn[name] = getattr(_lib or load(), "faoctasr_" + name)
    t = launch_timer
    if t is not None and name in t.names:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        t.add(name, args, s, e, _lib.faoctasr_last_route())
    else:
        rc = fn(*args)
    if rc != 0:
        raise KernelError("faoctasr_%s failed (%d): %s" % (name, rc, _lib.faoctasr_last_error().decode()))


_ws = {}


def workspace(device, nfloats, tag=None):
    """Per-(device, stream[, tag]) scratch reused by the stream-ordered reduction kernels."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream, tag)
    t = _ws.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _ws[key] = t
    return t
