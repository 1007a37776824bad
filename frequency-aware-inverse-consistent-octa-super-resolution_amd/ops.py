"""Autograd operators over the HIP kernels (include/faoctasr.h).

Each ``torch.autograd.Function`` here stands in for one ATen op class the reference's train
step executes; PyTorch supplies device memory, streams and the autograd tape,
every arithmetic kernel is ours.  Weight / affine gradients are accumulated by the kernels
straight into ``param.grad`` when that already exists (the flat gradient arena of
train.TrainStep), so no per-parameter add kernels run and the all-reduce sees one buffer.
"""
import contextlib
import ctypes
import math

import torch
from torch.autograd import Function

from . import _lib
from ._lib import call, ptr, stream_ptr

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
_ACT = {None: 0, "none": 0, "relu": 1, "lrelu": 2, "tanh": 3}

#: accumulate weight gradients in place into existing ``param.grad`` buffers
direct_grad = True
#: residual blocks hand the skip connection's gradient to their first convolution's input-gradient call (model._ResBlock,
#: faoctasr_conv_set_residual) instead of leaving the add to autograd; False: plain autograd (tests compare the two)
fuse_residual_grad = True


def act_code(a):
    return a if isinstance(a, int) else _ACT[a]


def _c(t):
    if not t.is_cuda:
        raise _lib.KernelError("operand lives on %s: the HIP operators have no CPU/eager fallback" % t.device)
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------
# packed-weight images for the LDS-patch convolution kernel (include/faoctasr.h, faoctasr_conv_wpack_floats)
# ----------------------------------------------------------------------------------------
import weakref

#: bumped whenever weights change behind autograd's back and nobody knows which (``invalidate_weight_cache``); a ``ParamArena``
#: bumps only its own parameters' epoch cell (``p._fa_epoch``), so a generator update leaves the discriminators' images valid
weight_epoch = 0
use_wpack = True
#: convolution arithmetic: 0 = fp32 MFMA with Winograd F(2x2,3x3) on the stride-1 3x3 layers (default), 1 = fp32 MFMA direct only,
#: 2 = bf16x3 split operands (fp32-parity, ~5x MFMA rate); 3 = f16x2 split operands with per-tensor power-of-two scales (error at
#: or below the exact-f32 kernels', same rate as bf16x3: csrc/split16.h)
conv_precision = 0
PRECISIONS = {"f32": 0, "f32_direct": 1, "bf16x3": 2, "f16x2": 3}
#: FAOCTASR_CONV_NO_SPLIT_K (include/faoctasr.h) on every FORWARD convolution: no fp32 atomics in the forward pass, so its
#: activations -- and with them every ReLU / LeakyReLU mask the backward uses -- are bit-reproducible from run to run
reproducible_forward = False
NO_SPLIT_K = 0x100
_wpack_cache = {}
#: calls that had to pack inside the convolution call itself (state 1); a ``PackPlan`` owner watches it to learn about new images
pack_misses = 0


def invalidate_weight_cache():
    global weight_epoch
    weight_epoch += 1


# ----------------------------------------------------------------------------------------
# absmax slots of the f16x2 contraction (include/faoctasr.h: faoctasr_absmax_bits, faoctasr_conv_set_scales)
# ----------------------------------------------------------------------------------------
#: per HIP stream: (zeroed fp32 arena, cursor).  A slot is SLOT_WORDS words whose maximum is max|x| of an activation tensor (the fp32
#: bit pattern IS the float).  One arena per stream, because a slot is zeroed, filled and first read in stream order -- handing slots of one
#: arena to several streams would let a reader overtake the fill.
_scale_arenas = {}
SCALE_ARENA_SLOTS = 1024
SLOT_WORDS = 128                                # include/faoctasr.h FAOCTASR_ABSMAX_SLOT_WORDS: 8 used words, one per 64-byte line


def reset_scale_arenas():
    """Drop the arenas (their slots stay alive while something references them): the next slot comes from a freshly zeroed
    arena.  ``GraphedTrainStep`` calls it around a capture so that every slot of the graph is zeroed INSIDE the graph."""
    _scale_arenas.clear()


absmax_log = None


def _new_slot(device):
    sid = stream_ptr()
    ent = _scale_arenas.get(sid)
    if ent is None or ent[1] >= SCALE_ARENA_SLOTS or ent[0].device != device:
        ent = _scale_arenas[sid] = [torch.zeros(SCALE_ARENA_SLOTS * SLOT_WORDS, dtype=torch.float32, device=device), 0]
    i = ent[1]
    ent[1] = i + 1
    return ent[0][i * SLOT_WORDS:(i + 1) * SLOT_WORDS]


def absmax_slot(t):
    """The absmax slot of activation tensor ``t`` for a convolution enqueued on the CURRENT stream.  Cached on the tensor object for
    further uses on the same stream while the tensor is unchanged (the skip connection and the first convolution of a residual
    block read the same tensor); a producer that computes the maximum in its own epilogue may pre-set ``t._fa_absmax``."""
    sid = stream_ptr()
    c = getattr(t, "_fa_absmax", None)
    if c is not None and c[1] == t._version and c[2] == sid:
        return c[0]
    slot = _new_slot(t.device)
    call("absmax_bits", ptr(t), t.numel(), ptr(slot), sid)
    t._fa_absmax = (slot, t._version, sid)
    if absmax_log is not None:                      # (diagnostics) which tensors still need a stand-alone pass, and why
        why = "no tag" if c is None else ("version" if c[1] != t._version else "stream")
        key = (tuple(t.shape), why)
        absmax_log[key] = absmax_log.get(key, 0) + 1
    return slot


def _producer_slot(x, C, H, W):
    """A fresh absmax slot for a producer call (BatchNorm forward / backward, cat2_act forward) whose output can feed a
    split-precision convolution (>= 16 channels, a map >= 24 wide, H*W a multiple of 4); None otherwise.  ``_producer_call`` hands it
    over."""
    if conv_precision != 3 or C < 16 or W < 24 or (H * W) & 3:
        return None
    return _new_slot(x.device)


def _producer_call(name, args, slot):
    """A producer call with ``faoctasr_out_absmax(slot)`` set immediately before it (``args`` are already evaluated: nothing that can
    raise sits between the setter and its consumer)."""
    if slot is not None:
        call("out_absmax", ptr(slot))
    call(name, *args)


def _tag_absmax(t, slot):
    """``t``'s producer has folded max|t| into ``slot`` on the current stream: ``absmax_slot(t)`` will find it."""
    if slot is not None:
        t._fa_absmax = (slot, t._version, stream_ptr())


_wgrad_ws_need = {}


def _wgrad_workspace(device, C, M, KH, KW, stride, prec):
    """Hand the CURRENT stream's scratch buffer to the next weight-gradient call (two-pass reduction of the split kernels:
    ``faoctasr_conv_set_workspace``).  Calls on one stream share the buffer; it is only live between the call's two launches."""
    if prec < 2:
        return
    key = (C, M, KH, KW, stride)
    n = _wgrad_ws_need.get(key)
    if n is None:
        n = _wgrad_ws_need[key] = _lib.load().faoctasr_conv_wgrad_workspace_floats(C, M, KH, KW, stride)
    if n > 0:
        ws = _lib.workspace(device, n + (1 << 16), tag="wgrad")
        call("conv_set_workspace", ptr(ws), ws.numel())


def _conv_call(name, args, slot_a=None, slot_b=None, residual=None):
    """One convolution-type C call with its hand-over state (``faoctasr_conv_set_scales`` / ``faoctasr_conv_set_residual``: thread-local,
    consumed by the next call) set IMMEDIATELY before it: nothing that can raise -- an allocation, a pack, a pointer check -- sits
    between a setter and the call that consumes it, so a failed step cannot leave a slot or a residual pointer behind for an unrelated
    later call."""
    pa, pb, pr = ptr(slot_a), ptr(slot_b), ptr(residual)
    if pa is not None:
        call("conv_set_scales", pa, pb)
    if pr is not None:
        call("conv_set_residual", pr)
    call(name, *args)


_scale_need = {}


def _needs_scales(kind, dims, reflect=0, out_pad=0):
    """Which absmax slots the precision-3 form of a convolution-type call reads (``faoctasr_conv_needs_scales``: answered by the
    dispatch code itself): 0 = none (an exact-f32 route), 1 = slot a, 3 = both.  Cached per call signature."""
    if conv_precision != 3:
        return 0
    key = (kind, dims, reflect, out_pad)
    n = _scale_need.get(key)
    if n is None:
        n = _lib.load().faoctasr_conv_needs_scales(kind, *dims, reflect, out_pad)
        if n < 0:
            raise _lib.KernelError("conv_needs_scales: " + _lib.load().faoctasr_last_error().decode())
        _scale_need[key] = n
    return n


class _PackEntry:
    """One packed-weight image: the buffer, the gather call it belongs to (exactly the C call's arguments) and the weight
    version it holds."""
    __slots__ = ("wref", "buf", "ver", "kind", "dims", "reflect", "out_pad", "precision", "touched", "event", "ev_stream")


def _ver(w):
    cell = getattr(w, "_fa_epoch", None)
    return (w._version, weight_epoch, 0 if cell is None else cell[0], w.data_ptr(), tuple(w.shape))


def _wpack(w, kind, dims, reflect=0, out_pad=0):
    """(buffer, state, entry) for the C ABI: state 1 = pack now, 2 = buffer already holds these weights.  ``dims`` =
    (N, C, IH, IW, M, KH, KW, stride, pad) exactly as the gather call ``kind`` receives them.

    An image packed inside a convolution call (state 1) is ordered on that call's stream only.  The caller reports the launch with
    ``_packed(entry, state)``, which records an event there; a later state-2 user on ANOTHER stream waits for that event here, so the
    multi-stream schedule of ``TrainStep`` is safe on steps whose images are not covered by a ``PackPlan`` (first step, new batch
    shape: the trailing partial batch of an epoch)."""
    global pack_misses
    if not use_wpack:
        return None, 0, None
    key = (id(w), kind, conv_precision, dims, reflect, out_pad)
    ent = _wpack_cache.get(key)
    if ent is None or ent.wref() is not w:
        if len(_wpack_cache) > 4096:
            for k in [k for k, v in _wpack_cache.items() if v.wref() is None]:
                del _wpack_cache[k]
        N, C, IH, IW, M, KH, KW, stride, pad = dims
        n = _lib.load().faoctasr_conv_wpack_floats(kind, C, M, KH, KW, stride, pad, conv_precision)
        if n <= 0:
            return None, 0, None
        ent = _PackEntry()
        ent.wref, ent.buf, ent.ver = weakref.ref(w), torch.empty(n, dtype=torch.float32, device=w.device), None
        ent.event = ent.ev_stream = None
        ent.kind, ent.dims, ent.reflect, ent.out_pad, ent.precision = kind, dims, int(bool(reflect)), out_pad, conv_precision
        _wpack_cache[key] = ent
    ver = _ver(w)
    ent.touched = True
    if ent.ver == ver:
        if ent.event is not None:
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != ent.ev_stream:         # packed inline on another stream: order this reader behind that launch
                if wait_guard is not None:
                    wait_guard(cur.cuda_stream, ent.ev_stream)
                cur.wait_event(ent.event)
        return ent.buf, 2, ent
    ent.ver = ver
    pack_misses += 1
    return ent.buf, 1, ent


def _packed(ent, state):
    """After a convolution call that received ``state`` 1: remember where (stream) and when (event) the image was written."""
    if state == 1:
        cur = torch.cuda.current_stream()
        ent.event, ent.ev_stream = cur.record_event(), cur.cuda_stream


def clear_touched():
    """Forget which images earlier steps used: the next ``PackPlan`` then holds exactly the images of the step run in between."""
    for e in _wpack_cache.values():
        e.touched = False


class PackPlan:
    """Every packed-weight image of a set of parameters in one launch (``faoctasr_conv_pack_job`` / ``faoctasr_conv_pack_run``,
    csrc/conv_pack.hip): the images the last step used (``touched``) at ``precision``, for weights in ``params``.  ``run()``
    packs them all and marks them current, so the convolution calls that follow pass ``wpack_state`` 2."""

    def __init__(self, params, precision):
        lib = _lib.load()
        ids = {id(p) for p in params}
        slot = ctypes.create_string_buffer(_lib.PACK_JOB_BYTES)
        blobs, base, self.entries, device = [], 0, [], None
        self.precision = precision
        for key, e in list(_wpack_cache.items()):
            w = e.wref()
            if w is None or id(w) not in ids or e.precision != precision or not e.touched:
                continue
            e.touched = False
            n = lib.faoctasr_conv_pack_job(slot, base, e.kind, w.data_ptr(), e.buf.data_ptr(), *e.dims, e.reflect, e.out_pad, e.precision)
            if n < 0:
                raise _lib.KernelError("conv_pack_job: " + lib.faoctasr_last_error().decode())
            self.entries.append((e, w.data_ptr()))     # n == 0: the call never reads its image (64->1 head): nothing to pack, still "current"
            if n > 0:
                blobs.append(slot.raw)
                base += n
                device = w.device
        self.njobs, self.nblocks = len(blobs), base
        self.table = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.float32).to(device) if blobs else None   # raw bytes, carried as fp32

    def run(self, aside=None):
        """False (nothing launched) when a weight has moved or died since the jobs were recorded: the owner drops the plan.

        ``aside``: a stream to pack on instead of the current one (ordered behind the current stream's work so far).  Readers on that
        stream see the images in stream order; readers on any other stream are ordered behind the launch by the event ``_wpack``
        already honours for images packed inside a convolution call."""
        live = [(e, e.wref()) for e, _ in self.entries]
        if any(w is None or w.data_ptr() != p for (e, w), (_, p) in zip(live, self.entries)):
            return False
        event = ev_stream = None
        if self.njobs:
            if aside is not None:
                cur = torch.cuda.current_stream()
                if cur.cuda_stream == aside.cuda_stream:
                    aside = None
                else:
                    aside.wait_stream(cur)
            with torch.cuda.stream(aside) if aside is not None else contextlib.nullcontext():
                if self.precision == 3:             # f16x2 images: the weights' absmax slots first
                    call("conv_pack_scales", ptr(self.table), self.njobs, stream_ptr())
                call("conv_pack_run", ptr(self.table), self.njobs, self.nblocks, stream_ptr())
                if aside is not None:
                    event, ev_stream = aside.record_event(), aside.cuda_stream
        for e, w in live:
            e.ver = _ver(w)
            e.event, e.ev_stream = event, ev_stream      # None: written on the stream every role of the step forks from
        return True


#: Set by TrainStep for the span of a backward pass: weight / bias gradients that accumulate straight into the gradient arena are
#: then enqueued on this second HIP stream (ordered after the producing stream's work so far), so that they overlap the
#: input-gradient chain, the BatchNorm backward passes and each other's launch tails instead of queueing behind them.
#: ``join_wgrad_stream`` makes the current stream wait for them (before the all-reduce / optimizer step).  Their operands stay
#: referenced until then, so the caching allocator cannot hand an activation to another kernel while a side-stream kernel still
#: reads it.
wgrad_stream = None
_wgrad_keep = {}
#: ``f(waiter_stream_id, waited_stream_id)`` called before each cross-stream wait issued here (train._CaptureGuard.edge)
wait_guard = None


#: (captured steps, TrainStep.capture_side_wgrad == "deferred") collect the weight-gradient launches of a backward pass and enqueue
#: them on the side stream in ONE batch when the pass is over (``end_wgrad``): one fork and one join per backward pass in the graph
#: instead of one per weight gradient
wgrad_defer = False
_deferred = []


def end_wgrad():
    """Close a ``wgrad_stream`` scope: enqueue what ``wgrad_defer`` held back, then reset the stream."""
    global wgrad_stream
    side, held = wgrad_stream, list(_deferred)
    del _deferred[:]
    wgrad_stream = None
    if held and side is not None:
        seen = {side.cuda_stream}
        for _, _, src in held + [(None, None, torch.cuda.current_stream())]:      # every stream a held launch's operands came from
            if src.cuda_stream not in seen:
                seen.add(src.cuda_stream)
                if wait_guard is not None:
                    wait_guard(side.cuda_stream, src.cuda_stream)
                side.wait_stream(src)
        with torch.cuda.stream(side):
            for fn, operands, _ in held:
                fn(side.cuda_stream)
                _wgrad_keep.setdefault(side.cuda_stream, []).append(operands)
    elif held:
        for fn, _, _ in held:
            fn(stream_ptr())


def _enqueue_wgrad(fn, *operands):
    """``fn(stream_pointer)`` on ``wgrad_stream`` when one is set, otherwise on the current stream."""
    side = wgrad_stream
    if side is None:
        fn(stream_ptr())
        return
    if wgrad_defer:
        _deferred.append((fn, operands, torch.cuda.current_stream()))
        return
    cur = torch.cuda.current_stream()
    if cur.cuda_stream != side.cuda_stream:
        if wait_guard is not None:
            wait_guard(side.cuda_stream, cur.cuda_stream)
        side.wait_stream(cur)
    with torch.cuda.stream(side):
        fn(side.cuda_stream)
    _wgrad_keep.setdefault(side.cuda_stream, []).append(operands)


def join_wgrad_stream(side):
    """The current stream waits for everything enqueued on ``side``; the operands held for it are released."""
    if side is not None:
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != side.cuda_stream:
            if wait_guard is not None:
                wait_guard(cur.cuda_stream, side.cuda_stream)
            cur.wait_stream(side)
        _wgrad_keep.pop(side.cuda_stream, None)


def _grad_target(p):
    g = p.grad
    if direct_grad and g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g
    return None


# ----------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------
class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, reflect, act, slope, link=None):
        """``link``: the dict a residual block shares between its first convolution and its last BatchNorm (``model._ResBlock``): the
        BatchNorm's backward leaves the skip connection's gradient there (``link["dres"]``) instead of returning it to autograd, and
        this convolution's input gradient is computed as dgrad(dy) + dres in one kernel (``faoctasr_conv_set_residual``)."""
        ctx.link = link
        x, w = _c(x), _c(w)
        N, C, IH, IW = x.shape
        M, Cw, KH, KW = w.shape
        if Cw != C:
            raise _lib.KernelError("conv2d: input has %d channels, weight expects %d" % (C, Cw))
        OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
        if OH <= 0 or OW <= 0:
            raise RuntimeError("Calculated padded input size per channel: (%d x %d). Kernel size: (%d x %d). "
                               "Kernel size can't be greater than actual input size" % (IH + 2 * pad, IW + 2 * pad, KH, KW))
        y = torch.empty((N, M, OH, OW), dtype=torch.float32, device=x.device)
        wp, wst, ent = _wpack(w, 0, (N, C, IH, IW, M, KH, KW, stride, pad), reflect)
        sx = absmax_slot(x) if _needs_scales(0, (N, C, IH, IW, M, KH, KW, stride, pad), reflect) else None
        _conv_call("conv2d_fwd", (ptr(x), ptr(w), ptr(bias), ptr(y), N, C, IH, IW, M, KH, KW, stride, pad, reflect, act, slope, ptr(wp), wst,
                                  conv_precision | (NO_SPLIT_K if reproducible_forward else 0), stream_ptr()), sx)
        _packed(ent, wst)
        ctx.sx = sx
        ctx.save_for_backward(x, w, y if act else None)
        ctx.w_ref, ctx.b_ref = w, bias
        ctx.cfg = (stride, pad, reflect, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, reflect, act, slope = ctx.cfg
        dy = _c(dy)
        st = stream_ptr()
        N, C, IH, IW = x.shape
        M, _, KH, KW = w.shape
        if act:
            g = torch.empty_like(dy)
            call("act_bwd", ptr(dy), ptr(y), ptr(g), dy.numel(), act, slope, st)
            dy = g
        dx = dw = db = None
        dg_dims = (N, C, IH + 2 * pad, IW + 2 * pad, M, KH, KW, stride, 0) if reflect else (N, C, IH, IW, M, KH, KW, stride, pad)
        need_d = _needs_scales(1, dg_dims) if ctx.needs_input_grad[0] else 0
        need_w = _needs_scales(4, (N, C, IH, IW, M, KH, KW, stride, pad), reflect) if ctx.needs_input_grad[1] else 0
        sdy = absmax_slot(dy) if (need_d or need_w) else None
        res = ctx.link.pop("dres", None) if ctx.link is not None else None
        if res is not None and not ctx.needs_input_grad[0]:
            raise _lib.KernelError("a residual link left a skip gradient for a convolution whose input needs no gradient")
        if ctx.needs_input_grad[0]:
            sd = sdy if need_d else None
            if res is not None:
                res = _c(res)
            if reflect:
                dxp = torch.empty((N, C, IH + 2 * pad, IW + 2 * pad), dtype=torch.float32, device=x.device)
                wp, wst, ent = _wpack(ctx.w_ref, 1, (N, C, IH + 2 * pad, IW + 2 * pad, M, KH, KW, stride, 0))
                _conv_call("conv2d_dgrad", (ptr(dy), ptr(w), ptr(dxp), N, C, IH + 2 * pad, IW + 2 * pad, M, KH, KW, stride, 0, ptr(wp), wst,
                                            conv_precision, st), sd)
                _packed(ent, wst)
                dx = torch.empty_like(x)
                call("reflect_pad_bwd", ptr(dxp), ptr(dx), N * C, IH, IW, pad, st)
                if res is not None:
                    call("axpby", ptr(dx), ptr(res), ptr(dx), dx.numel(), 1.0, 1.0, st)
            else:
                dx = torch.empty_like(x)
                wp, wst, ent = _wpack(ctx.w_ref, 1, (N, C, IH, IW, M, KH, KW, stride, pad))
                _conv_call("conv2d_dgrad", (ptr(dy), ptr(w), ptr(dx), N, C, IH, IW, M, KH, KW, stride, pad, ptr(wp), wst, conv_precision, st), sd, None, res)
                _packed(ent, wst)
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(ctx.w_ref)
            if tgt is None:
                dw = torch.empty_like(w)
            prec = conv_precision
            sx = None
            if need_w:
                sx = ctx.sx if ctx.sx is not None else absmax_slot(x)
            elif prec == 3:
                prec = 0                               # a shape the split weight-gradient kernels leave to the exact-f32 ones: no slots
            sdw = sdy if need_w else None

            def wgrad(s, out, accumulate):
                _wgrad_workspace(x.device, C, M, KH, KW, stride, prec)
                _conv_call("conv2d_wgrad", (ptr(x), ptr(dy), ptr(out), N, C, IH, IW, M, KH, KW, stride, pad, reflect, accumulate, prec, s), sx, sdw)
            if tgt is not None:
                _enqueue_wgrad(lambda s: wgrad(s, tgt, 1), x, dy, sx, sdw)
            else:
                wgrad(st, dw, 0)
        if ctx.b_ref is not None and ctx.needs_input_grad[2]:
            tb = _grad_target(ctx.b_ref)
            if tb is not None:
                _enqueue_wgrad(lambda s: call("channel_sum", ptr(dy), ptr(tb), N, M, dy.shape[2] * dy.shape[3], 1, s), dy)
            else:
                db = torch.empty(M, dtype=torch.float32, device=x.device)
                call("channel_sum", ptr(dy), ptr(db), N, M, dy.shape[2] * dy.shape[3], 0, st)
        return dx, dw, db, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, reflect=False, act=None, slope=0.2, link=None):
    return _Conv2d.apply(x, w, bias, int(stride), int(pad), 1 if reflect else 0, act_code(act), float(slope), link)


class _ConvTranspose2d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, out_pad, act, slope):
        x, w = _c(x), _c(w)
        N, C, IH, IW = x.shape
        Cw, M, KH, KW = w.shape
        if Cw != C:
            raise _lib.KernelError("conv_transpose2d: input has %d channels, weight expects %d" % (C, Cw))
        OH, OW = (IH - 1) * stride - 2 * pad + KH + out_pad, (IW - 1) * stride - 2 * pad + KW + out_pad
        y = torch.empty((N, M, OH, OW), dtype=torch.float32, device=x.device)
        wp, wst, ent = _wpack(w, 2, (N, C, IH, IW, M, KH, KW, stride, pad), 0, out_pad)
        sx = absmax_slot(x) if _needs_scales(2, (N, C, IH, IW, M, KH, KW, stride, pad), 0, out_pad) else None
        ctx.sx = sx
        _conv_call("conv_transpose2d_fwd", (ptr(x), ptr(w), ptr(bias), ptr(y), N, C, IH, IW, M, KH, KW, stride, pad, out_pad, act, slope,
                                            ptr(wp), wst, conv_precision | (NO_SPLIT_K if reproducible_forward else 0), stream_ptr()), sx)
        _packed(ent, wst)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.w_ref, ctx.b_ref = w, bias
        ctx.cfg = (stride, pad, out_pad, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, out_pad, act, slope = ctx.cfg
        dy = _c(dy)
        st = stream_ptr()
        N, C, IH, IW = x.shape
        _, M, KH, KW = w.shape
        if act:
            g = torch.empty_like(dy)
            call("act_bwd", ptr(dy), ptr(y), ptr(g), dy.numel(), act, slope, st)
            dy = g
        dx = dw = db = None
        need_d = _needs_scales(3, (N, C, IH, IW, M, KH, KW, stride, pad), 0, out_pad) if ctx.needs_input_grad[0] else 0
        need_w = _needs_scales(5, (N, C, IH, IW, M, KH, KW, stride, pad), 0, out_pad) if ctx.needs_input_grad[1] else 0
        sdy = absmax_slot(dy) if (need_d or need_w) else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wp, wst, ent = _wpack(ctx.w_ref, 3, (N, C, IH, IW, M, KH, KW, stride, pad), 0, out_pad)
            _conv_call("conv_transpose2d_dgrad", (ptr(dy), ptr(w), ptr(dx), N, C, IH, IW, M, KH, KW, stride, pad, out_pad, ptr(wp), wst,
                                                  conv_precision, st), sdy if need_d else None)
            _packed(ent, wst)
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(ctx.w_ref)
            if tgt is None:
                dw = torch.empty_like(w)
            prec = conv_precision
            sx = None
            if need_w:
                sx = ctx.sx if ctx.sx is not None else absmax_slot(x)
            elif prec == 3:
                prec = 0
            sdw = sdy if need_w else None

            def wgrad(s, out, accumulate):
                _wgrad_workspace(x.device, M, C, KH, KW, stride, prec)     # (the kernel sees x and dy swapped)
                _conv_call("conv_transpose2d_wgrad", (ptr(x), ptr(dy), ptr(out), N, C, IH, IW, M, KH, KW, stride, pad, out_pad, accumulate, prec, s),
                           sx, sdw)
            if tgt is not None:
                _enqueue_wgrad(lambda s: wgrad(s, tgt, 1), x, dy, sx, sdw)
            else:
                wgrad(st, dw, 0)
        if ctx.b_ref is not None and ctx.needs_input_grad[2]:
            tb = _grad_target(ctx.b_ref)
            if tb is not None:
                _enqueue_wgrad(lambda s: call("channel_sum", ptr(dy), ptr(tb), N, M, dy.shape[2] * dy.shape[3], 1, s), dy)
            else:
                db = torch.empty(M, dtype=torch.float32, device=x.device)
                call("channel_sum", ptr(dy), ptr(db), N, M, dy.shape[2] * dy.shape[3], 0, st)
        return dx, dw, db, None, None, None, None, None


def conv_transpose2d(x, w, bias=None, stride=1, pad=0, out_pad=0, act=None, slope=0.2):
    return _ConvTranspose2d.apply(x, w, bias, int(stride), int(pad), int(out_pad), act_code(act), float(slope))


# ----------------------------------------------------------------------------------------
# BatchNorm2d (training statistics) + activation + residual
# ----------------------------------------------------------------------------------------
class _BatchNormTrain(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, act, slope, link=None):
        ctx.link = link
        x = _c(x)
        N, C, H, W = x.shape
        if N * H * W <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        if residual is not None:
            residual = _c(residual)
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(x.device, C * 128)
        sp = stats.data_ptr()                   # (row views cost ~3 us each on the host: 243 calls per step)
        slot = _producer_slot(x, C, H, W)       # f16x2: the output's largest magnitude comes out of this kernel's store loop
        _producer_call("batchnorm_train_fwd", (ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(y), sp, sp + 4 * C,
                                               ptr(running_mean), ptr(running_var), N, C, H * W, eps, momentum, act, slope, ptr(ws), stream_ptr()), slot)
        # the backward takes the ReLU / LeakyReLU mask from x when there was no residual (faoctasr.h): y is then not kept by this node
        need_y = act == ACT_TANH or (act and residual is not None)
        ctx.save_for_backward(x, gamma, stats, y if need_y else None, beta if (act and not need_y) else None)
        ctx.g_ref, ctx.b_ref = gamma, beta
        ctx.cfg = (act, slope, residual is not None)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensor for `stats` on every backward
        _tag_absmax(y, slot)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, gamma, stats, y, beta = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        dy = _c(dy)
        N, C, H, W = x.shape
        dx = torch.empty_like(x)
        need_affine = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        tg = tb = dgamma = dbeta = None
        accumulate = 0
        if need_affine:
            tg, tb = _grad_target(ctx.g_ref), _grad_target(ctx.b_ref)
            if tg is not None and tb is not None:
                accumulate = 1
            else:
                tg = tb = None
                dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
                dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        dres = None
        if has_res and ctx.needs_input_grad[3]:
            dres = torch.empty_like(x) if act else dy
        ws = _lib.workspace(x.device, C * 128)
        sp = stats.data_ptr()
        slot = _producer_slot(x, C, H, W)       # f16x2: dx is the dY of the convolution in front of this layer
        _producer_call("batchnorm_train_bwd", (ptr(x), ptr(dy), ptr(y), ptr(gamma), ptr(beta), sp, sp + 4 * C, ptr(dx),
                                               ptr(tg if accumulate else dgamma), ptr(tb if accumulate else dbeta),
                                               ptr(dres) if (dres is not None and act) else None,
                                               N, C, H * W, act, slope, accumulate, ptr(ws), stream_ptr()), slot)
        _tag_absmax(dx, slot)
        if ctx.link is not None and dres is not None:
            ctx.link["dres"] = dres          # the block's first convolution adds it to its input gradient in its own epilogue
            dres = None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


class _BatchNormEval(Function):
    """Inference-mode BatchNorm2d (running statistics): y = act(gamma * (x - mean) / sqrt(var + eps) + beta)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, act, slope):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        call("batchnorm_eval_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(y), N, C, H * W, eps, act, slope,
             stream_ptr())
        ctx.save_for_backward(gamma, running_var, y if act else None)
        ctx.cfg = (eps, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        gamma, rv, y = ctx.saved_tensors
        eps, act, slope = ctx.cfg
        dy = _c(dy)
        N, C, H, W = dy.shape
        dx = torch.empty_like(dy)
        call("batchnorm_eval_bwd", ptr(dy), ptr(y), ptr(gamma), ptr(rv), ptr(dx), N, C, H * W, eps, act, slope, stream_ptr())
        return dx, None, None, None, None, None, None, None


def batchnorm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, act=None, slope=0.2):
    return _BatchNormEval.apply(x, gamma, beta, running_mean, running_var, float(eps), act_code(act), float(slope))


def batchnorm_train(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, act=None, slope=0.2, residual=None, link=None):
    y, _ = _BatchNormTrain.apply(x, gamma, beta, residual, running_mean, running_var, float(momentum), float(eps), act_code(act), float(slope),
                                 link if residual is not None else None)
    return y


class _InstanceNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, slope):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((2, N * C), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(x.device, N * C * 128)
        call("instancenorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats[0]), ptr(stats[1]), N, C, H * W, eps, act, slope,
             ptr(ws), stream_ptr())
        ctx.save_for_backward(x, gamma, stats, y if act == ACT_TANH else None, beta if act and act != ACT_TANH else None)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats, y, beta = ctx.saved_tensors
        act, slope = ctx.cfg
        dy = _c(dy)
        N, C, H, W = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        ws = _lib.workspace(x.device, N * C * 128)
        call("instancenorm_bwd", ptr(x), ptr(dy), ptr(y), ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]), ptr(dx), ptr(dgamma), ptr(dbeta),
             N, C, H * W, act, slope, ptr(ws), stream_ptr())
        return dx, dgamma, dbeta, None, None, None


def instance_norm(x, gamma=None, beta=None, eps=1e-5, act=None, slope=0.2):
    return _InstanceNorm.apply(x, gamma, beta, float(eps), act_code(act), float(slope))


# ----------------------------------------------------------------------------------------
# pointwise
# ----------------------------------------------------------------------------------------
class _Act(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _c(x)
        y = torch.empty_like(x)
        call("act_fwd", ptr(x), ptr(y), x.numel(), act, slope, stream_ptr())
        ctx.save_for_backward(y)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        call("act_bwd", ptr(dy), ptr(y), ptr(dx), dy.numel(), ctx.cfg[0], ctx.cfg[1], stream_ptr())
        return dx, None, None


def activation(x, act, slope=0.2):
    return _Act.apply(x, act_code(act), float(slope))


class _Cat2Act(Function):
    @staticmethod
    def forward(ctx, a, b, act, slope):
        a, b = _c(a), _c(b)
        N, Ca, H, W = a.shape
        Cb = b.shape[1]
        y = torch.empty((N, Ca + Cb, H, W), dtype=torch.float32, device=a.device)
        slot = _producer_slot(a, Ca + Cb, H, W)
        _producer_call("cat2_act_fwd", (ptr(a), ptr(b), ptr(y), N, Ca, Cb, H * W, act, slope, stream_ptr()), slot)
        _tag_absmax(y, slot)
        ctx.save_for_backward(y if act else None)
        ctx.cfg = (act, slope, Ca, Cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope, Ca, Cb = ctx.cfg
        dy = _c(dy)
        N, _, H, W = dy.shape
        da = torch.empty((N, Ca, H, W), dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        db = torch.empty((N, Cb, H, W), dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
        call("cat2_act_bwd", ptr(dy), ptr(y), ptr(da), ptr(db), N, Ca, Cb, H * W, act, slope, stream_ptr())
        return da, db, None, None


def cat2_act(a, b, act=None, slope=0.2):
    """torch.cat([a, b], 1) followed by an activation (model.py:266+249, 268+431, 298+431)."""
    return _Cat2Act.apply(a, b, act_code(act), float(slope))


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        call("axpby", ptr(a), ptr(b), ptr(y), a.numel(), 1.0, 1.0, stream_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _Axpby(Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        call("axpby", ptr(a), ptr(b), ptr(y), a.numel(), alpha, beta, stream_ptr())
        ctx.cfg = (alpha, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        alpha, beta = ctx.cfg
        dy = _c(dy)
        st = stream_ptr()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(dy)
            call("axpby", ptr(dy), ptr(dy), ptr(da), dy.numel(), alpha, 0.0, st)
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(dy)
            call("axpby", ptr(dy), ptr(dy), ptr(db), dy.numel(), beta, 0.0, st)
        return da, db, None, None


def axpby(a, b, alpha, beta):
    return _Axpby.apply(a, b, float(alpha), float(beta))


# ----------------------------------------------------------------------------------------
# Haar DWT
# ----------------------------------------------------------------------------------------
class _HaarAFB2D(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, C, H, W = x.shape
        ll = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        hi = torch.empty((N, C, 3, H // 2, W // 2), dtype=torch.float32, device=x.device)
        call("haar_dwt2d_fwd", ptr(x), ptr(ll), ptr(hi), N * C, H, W, stream_ptr())
        ctx.shape = (N, C, H, W)
        return ll, hi

    @staticmethod
    def backward(ctx, dll, dhi):
        N, C, H, W = ctx.shape
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=(dll if dll is not None else dhi).device)
        call("haar_dwt2d_bwd", ptr(_c(dll)) if dll is not None else None, ptr(_c(dhi)) if dhi is not None else None, ptr(dx), N * C, H, W,
             stream_ptr())
        return dx


class _HaarSFB2D(Function):
    @staticmethod
    def forward(ctx, ll, hi):
        ll, hi = _c(ll), _c(hi)
        N, C, h, w = ll.shape
        x = torch.empty((N, C, 2 * h, 2 * w), dtype=torch.float32, device=ll.device)
        call("haar_dwt2d_bwd", ptr(ll), ptr(hi), ptr(x), N * C, 2 * h, 2 * w, stream_ptr())
        return x

    @staticmethod
    def backward(ctx, dx):
        dx = _c(dx)
        N, C, H, W = dx.shape
        dll = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=dx.device)
        dhi = torch.empty((N, C, 3, H // 2, W // 2), dtype=torch.float32, device=dx.device)
        call("haar_dwt2d_fwd", ptr(dx), ptr(dll), ptr(dhi), N * C, H, W, stream_ptr())
        return dll, dhi


def haar_afb2d(x):
    return _HaarAFB2D.apply(x)


def haar_sfb2d(ll, hi):
    return _HaarSFB2D.apply(ll, hi)


class _HaarDFront(Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = _c(x)
        N, C, H, W = x.shape
        if C != 1:
            raise _lib.KernelError("discriminator front end expects single-channel images")
        y = torch.empty((N, 3 if mode else 1, H // 2, W // 2), dtype=torch.float32, device=x.device)
        call("haar_dfront_fwd", ptr(x), ptr(y), N, H, W, mode, stream_ptr())
        ctx.cfg = (N, H, W, mode)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, mode = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((N, 1, H, W), dtype=torch.float32, device=dy.device)
        call("haar_dfront_bwd", ptr(dy), ptr(dx), N, H, W, mode, stream_ptr())
        return dx, None


def haar_dfront(x, mode):
    """mode 0: LL (FS_DiscriminatorA, model.py:171-172); 1: cat(LH,HL,HH)*0.5+0.5 (FS_DiscriminatorB, model.py:225-233)."""
    return _HaarDFront.apply(x, int(mode))


# ----------------------------------------------------------------------------------------
# FFT Gaussian split as circulant GEMMs
# ----------------------------------------------------------------------------------------
_circ_cache = {}


def circulant_lowpass(n, radius, device):
    """Real symmetric circulant C with C x == ifft(ifftshift(g) * fft(x)) for the centred Gaussian taps
    g[k] = exp(-(k - int(n/2))^2 / (2 r^2)) of utils.py:71-80 (the 2-D mask is the outer product g g^T).  Built on the device
    by ``faoctasr_circulant_lowpass`` once per (n, radius, device); this dict is the caller-owned filter cache of SURVEY 8b."""
    device = torch.device(device)
    key = (n, float(radius), device.type, device.index if device.index is not None else torch.cuda.current_device())
    c = _circ_cache.get(key)
    if c is None:
        if device.type != "cuda":
            raise _lib.KernelError("circulant_lowpass: the filter matrices are built by a HIP kernel; got device %s" % device)
        c = torch.empty((n, n), dtype=torch.float32, device=device)
        call("circulant_lowpass", ptr(c), n, float(radius), stream_ptr())
        _circ_cache[key] = c
    return c


def _lowpass2d(x3, ch, cw, st):
    """x3: (B,H,W) -> Ch @ x_b @ Cw for every b (two MFMA SGEMM launches)."""
    B, H, W = x3.shape
    t = torch.empty_like(x3)
    call("sgemm_batched", ptr(x3), ptr(cw), ptr(t), B * H, W, W, W, W, W, 0, 0, 0, 1, st)
    y = torch.empty_like(x3)
    call("sgemm_batched", ptr(ch), ptr(t), ptr(y), H, W, H, H, W, W, 0, H * W, H * W, B, st)
    return y


class _FreqSplit(Function):
    @staticmethod
    def forward(ctx, x, r_hp, r_lp):
        x = _c(x)
        B, C, H, W = x.shape
        dev = x.device
        mats = (circulant_lowpass(H, r_hp, dev), circulant_lowpass(W, r_hp, dev), circulant_lowpass(H, r_lp, dev), circulant_lowpass(W, r_lp, dev))
        st = stream_ptr()
        x3 = x.view(B * C, H, W)
        low_hp = _lowpass2d(x3, mats[0], mats[1], st)
        low_lp = _lowpass2d(x3, mats[2], mats[3], st)
        hf, lf = torch.empty_like(x), torch.empty_like(x)
        call("freq_mix_fwd", ptr(x), ptr(low_hp), ptr(low_lp), ptr(hf), ptr(lf), x.numel(), st)
        ctx.save_for_backward(x, low_hp, low_lp)
        ctx.mats = mats
        return hf, lf

    @staticmethod
    def backward(ctx, g_hf, g_lf):
        x, low_hp, low_lp = ctx.saved_tensors
        mats = ctx.mats
        st = stream_ptr()
        B, C, H, W = x.shape
        s_hp, s_lp, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        call("freq_mix_bwd", ptr(x), ptr(low_hp), ptr(low_lp), ptr(_c(g_hf)) if g_hf is not None else None,
             ptr(_c(g_lf)) if g_lf is not None else None, ptr(s_hp), ptr(s_lp), ptr(dx), x.numel(), st)
        # the circulants are symmetric, so the adjoint of x -> Ch x Cw is the same map
        a = _lowpass2d(s_hp.view(B * C, H, W), mats[0], mats[1], st)
        b = _lowpass2d(s_lp.view(B * C, H, W), mats[2], mats[3], st)
        n = x.numel()
        call("axpby", ptr(dx), ptr(a), ptr(dx), n, 1.0, -1.0, st)
        call("axpby", ptr(dx), ptr(b), ptr(dx), n, 1.0, 1.0, st)
        return dx, None, None


def freq_split(x, r_hp, r_lp):
    """hf = (high_pass(x_b, r_hp) + x_b)/2, lf = low_pass(x_b, r_lp) per sample (train.py:173-175)."""
    return _FreqSplit.apply(x, float(r_hp), float(r_lp))


# ----------------------------------------------------------------------------------------
# losses and the discriminator head
# ----------------------------------------------------------------------------------------
LOSS_MSE, LOSS_L1, LOSS_BCE_LOGITS = 0, 1, 2


class _Loss(Function):
    @staticmethod
    def forward(ctx, a, b, kind, scale):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise _lib.KernelError("loss operands differ in shape: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _lib.workspace(a.device, 1024)
        call("loss_fwd", ptr(a), ptr(b), ptr(out), a.numel(), kind, scale, ptr(ws), stream_ptr())
        ctx.save_for_backward(a, b)
        ctx.cfg = (kind, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        kind, scale = ctx.cfg
        g = _c(g)
        da = db = None
        st = stream_ptr()
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            call("loss_bwd", ptr(a), ptr(b), ptr(g), ptr(da), a.numel(), kind, scale, 0, st)
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            call("loss_bwd", ptr(a), ptr(b), ptr(g), ptr(db), a.numel(), kind, scale, 1, st)
        return da, db, None, None


def mse_loss(a, b, weight=1.0):
    return _Loss.apply(a, b, LOSS_MSE, float(weight) / a.numel())


def l1_loss(a, b, weight=1.0):
    return _Loss.apply(a, b, LOSS_L1, float(weight) / a.numel())


def bce_with_logits(inp, target, weight=1.0):
    """torch.nn.BCEWithLogitsLoss()(inp, target); the gradient w.r.t. ``target`` is -inp/N (train.py:230-231)."""
    return _Loss.apply(inp, target, LOSS_BCE_LOGITS, float(weight) / inp.numel())


class _MeanMix(Function):
    @staticmethod
    def forward(ctx, a, b, wa, wb):
        a, b = _c(a), _c(b)
        N = a.shape[0]
        La, Lb = a.numel() // N, b.numel() // N
        out = torch.empty(N, dtype=torch.float32, device=a.device)
        call("mean_mix_fwd", ptr(a), ptr(b), ptr(out), N, La, Lb, wa, wb, stream_ptr())
        ctx.cfg = (a.shape, b.shape, wa, wb)
        return out

    @staticmethod
    def backward(ctx, g):
        sa, sb, wa, wb = ctx.cfg
        g = _c(g)
        N = sa[0]
        da = torch.empty(sa, dtype=torch.float32, device=g.device)
        db = torch.empty(sb, dtype=torch.float32, device=g.device)
        call("mean_mix_bwd", ptr(g), ptr(da), ptr(db), N, da.numel() // N, db.numel() // N, wa, wb, stream_ptr())
        return da, db, None, None


def mean_mix(a, b, wa=0.7, wb=0.3):
    """flatten(wa * global_avg_pool(a) + wb * global_avg_pool(b)) -- model.py:158-164."""
    return _MeanMix.apply(a, b, float(wa), float(wb))


# ----------------------------------------------------------------------------------------
# SSIM
# ----------------------------------------------------------------------------------------
class _SSIM(Function):
    @staticmethod
    def forward(ctx, a, b, size_average):
        a, b = _c(a), _c(b)
        N, C, H, W = a.shape
        sums = torch.empty(N, dtype=torch.float32, device=a.device)
        call("ssim_fwd", ptr(a), ptr(b), ptr(sums), N, C, H, W, stream_ptr())
        ctx.save_for_backward(a, b)
        ctx.size_average = size_average
        # the final division of N (or 1) numbers is host-side plumbing on a tiny tensor
        return sums.sum() / (N * C * H * W) if size_average else sums / (C * H * W)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        N, C, H, W = a.shape
        g = _c(g).reshape(-1)
        scale = 1.0 / (N * C * H * W) if ctx.size_average else 1.0 / (C * H * W)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        call("ssim_bwd", ptr(a), ptr(b), ptr(g), g.numel(), scale, ptr(da), ptr(db), N, C, H, W, stream_ptr())
        return da, db, None


def ssim(a, b, size_average=True):
    return _SSIM.apply(a, b, bool(size_average))
