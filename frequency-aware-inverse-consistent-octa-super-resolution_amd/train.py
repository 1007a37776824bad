"""The train step of the reference (train.py:166-269) as a driver over the HIP operators.

``TrainStep`` owns what the reference script builds at module level (train.py:73-126): the four
networks, the two AdamW optimizers (as two flat fp32 arenas: parameters, gradients and both
moments of every *live* tensor are contiguous, so the optimizer is one kernel launch and the
data-parallel exchange is one all-reduce per phase), the targets and the replay buffers.
``step(real_A, real_B)`` is one loop-body iteration.  The only semantic extension is the batched
per-sample frequency split (the reference supports batch 1 only: SURVEY.md fact 3).
"""
import ctypes
import math
import os
import random
import struct

import torch
import torch.distributed as dist

from . import ops
from ._lib import KernelError, call, ptr, stream_ptr
from .model import FS_DiscriminatorA, FS_DiscriminatorB, NetworkA2B, NetworkB2A
from .utils import DeviceReplayBuffer, ReplayBuffer, set_requires_grad, weights_init_normal
from .wavelets import DWTForward

# parameters that exist in the reference's state_dict but never receive a gradient
# (model.py:241,254-257: unet/unet_up of NetworkA2B; model.py:281-284: skip of NetworkB2A);
# torch.optim.AdamW skips them because their .grad stays None
DEAD_PREFIXES = {"NetworkA2B": ("unet.", "unet_up."), "NetworkB2A": ("skip.",)}


def live_parameters(net):
    dead = DEAD_PREFIXES.get(type(net).__name__, ())
    return [(n, p) for n, p in net.named_parameters() if not n.startswith(dead)]


class GradComm:
    """RCCL communicator behind the C ABI (``faoctasr_comm_*``, ``faoctasr_grad_allreduce``): one per process, bound to the
    current device.  The 128-byte RCCL id travels from group rank 0 over the already-initialised ``torch.distributed`` group
    (host channel only; the gradient bytes themselves never go through torch)."""

    def __init__(self, group=None):
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        ident = (ctypes.c_char * 128)()
        if self.rank == 0:
            call("comm_unique_id", ident)
        box = [bytes(ident.raw) if self.rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        handle = ctypes.c_void_p()
        call("comm_create", ctypes.byref(handle), self.world, self.rank, box[0])
        self.handle = handle
        self.ranks = _lib_int("comm_size", handle)
        if self.ranks != self.world:
            raise KernelError("RCCL communicator spans %d ranks, the process group %d" % (self.ranks, self.world))

    def all_reduce(self, t):
        call("grad_allreduce", ptr(t), t.numel(), 0, self.handle, stream_ptr())

    def broadcast(self, t, root=0):
        call("param_broadcast", ptr(t), t.numel(), root, self.handle, stream_ptr())

    def close(self):
        if self.handle:
            torch.cuda.synchronize()
            call("comm_destroy", self.handle)
            self.handle = None


def _lib_int(name, *args):
    """C-ABI queries that return a count (negative = error code)."""
    from . import _lib
    lib = _lib.load()
    v = getattr(lib, "faoctasr_" + name)(*args)
    if v < 0:
        raise KernelError("faoctasr_%s failed (%d): %s" % (name, v, lib.faoctasr_last_error().decode()))
    return v


class ParamArena:
    """Flat fp32 storage for a parameter group: ``param.data`` and ``param.grad`` become views of two
    contiguous buffers; AdamW moments live beside them.  Layout: tensors in registration order, each
    start aligned to 4 floats (16 B) so every view is float4-addressable."""

    def __init__(self, named_params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel = total
        self.live_numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.step_count = 0
        self._epoch = [0]                 # ops._ver: bumped by step(), which changes these parameters through raw pointers
        for p in self.params:
            p._fa_epoch = self._epoch

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def all_reduce(self, group=None, comm=None):
        """Sum gradients over ranks; the 1/world average is folded into the AdamW kernel's grad_scale.  With a ``GradComm``
        this is ``faoctasr_grad_allreduce`` (RCCL over xGMI, enqueued on the current stream, capturable); host arenas (the
        gloo CPU tests) go through ``torch.distributed``."""
        if comm is not None:
            comm.all_reduce(self.grad)
        else:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def step(self, grad_scale=1.0, hyper=None):
        """``hyper``: device tensor of 8 floats (``hyper_values``) -- the captured-graph form, whose scalars are read on the device."""
        if hyper is None:
            self.step_count += 1
            call("adamw_step", ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel, self.lr, self.betas[0],
                 self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale, stream_ptr())
        else:
            call("adamw_step_dev", ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel, ptr(hyper), stream_ptr())
        self._epoch[0] += 1                # the kernel changed these weights through raw pointers: their packed images are stale

    def hyper_values(self, step, grad_scale=1.0):
        """The 8 scalars of ``faoctasr_adamw_step_dev`` for optimizer step ``step`` (>= 1), rounded exactly as ``faoctasr_adamw_step``
        rounds its float arguments."""
        f32 = lambda v: struct.unpack("f", struct.pack("f", v))[0]
        lr, b1, b2 = f32(self.lr), f32(self.betas[0]), f32(self.betas[1])
        bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
        return [lr, b1, b2, f32(self.eps), f32(self.weight_decay), f32(lr / bc1), f32(1.0 / math.sqrt(bc2)), f32(grad_scale)]


class _CaptureGuard:
    """Host-side record of the waits issued between FORKED streams while a hipGraph capture is open.

    ROCm 7.2's ``hip::Stream::EndCapture`` walks each forked stream's list of "parallel capture streams" (the streams that waited
    on one of its events) recursively and without a visited set: a cycle among those waits -- two forked streams that wait on each
    other -- recurses until the stack overflows and the process dies with SIGSEGV in ``hipStreamEndCapture`` (DESIGN.md 4.4a,
    profiles/r03_capture_crash_gdb_*.log, tools/probe/capture_fork_join.hip).  Waits to and from the capture's origin stream are not
    listed.  ``edge`` is called before every cross-stream wait of the schedule; a wait that would close a cycle raises
    ``KernelError`` instead of being issued, so a ``stream_layout`` or schedule change that breaks the rule fails as a Python
    exception at capture time, not as a crash at the end of it."""

    def __init__(self):
        self.origin = None
        self.edges = {}

    def begin(self, origin_sid):
        self.origin, self.edges = origin_sid, {}

    def end(self):
        self.origin, self.edges = None, {}

    def edge(self, waiter_sid, on_sid):
        """Record "waiter waits on an event of on"; raises if that closes a cycle among forked streams."""
        if self.origin is None or waiter_sid == on_sid or waiter_sid == self.origin or on_sid == self.origin:
            return
        stack, seen = [on_sid], {on_sid}
        while stack:                                   # does `on` already (transitively) wait on `waiter`?
            n = stack.pop()
            if n == waiter_sid:
                raise KernelError("hipGraph capture: stream %#x would wait on stream %#x, which already waits on it -- a cycle of waits among "
                                  "forked capture streams crashes hipStreamEndCapture on ROCm 7.2 (DESIGN.md 4.4a); keep the role both depend "
                                  "on on the capture's origin stream" % (waiter_sid, on_sid))
            for m in self.edges.get(n, ()):
                if m not in seen:
                    seen.add(m)
                    stack.append(m)
        self.edges.setdefault(waiter_sid, set()).add(on_sid)


_guard = _CaptureGuard()
ops.wait_guard = _guard.edge          # the weight-gradient side stream's waits (ops._enqueue_wgrad / join_wgrad_stream) report here too


def _wait(waiter, on):
    """``waiter.wait_stream(on)`` unless both roles are the same stream (``TrainStep.stream_layout``).  A stream waiting on its own
    event is a no-op eagerly; inside a hipGraph capture it is not harmless on ROCm 7.2 (tools/probe/capture_fork_join.hip,
    profiles/r03_capture_probe.log), and it is never needed, so it is never issued."""
    if waiter.cuda_stream != on.cuda_stream:
        _guard.edge(waiter.cuda_stream, on.cuda_stream)
        waiter.wait_stream(on)


def _join(waiter, streams):
    """``waiter`` waits once for each distinct stream of ``streams`` (roles may share streams; never for itself)."""
    seen = {waiter.cuda_stream}
    for st in streams:
        if st is not None and st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            _guard.edge(waiter.cuda_stream, st.cuda_stream)
            waiter.wait_stream(st)


class _Mark:
    """An event together with the stream it was recorded on, so that a role sharing that stream can skip the wait (see ``_wait``)."""
    __slots__ = ("event", "sid")

    def __init__(self, stream):
        self.event, self.sid = stream.record_event(), stream.cuda_stream


def _after(waiter, mark):
    """``waiter`` continues after ``mark`` (no-op for None or a mark of the same stream)."""
    if mark is not None and waiter.cuda_stream != mark.sid:
        _guard.edge(waiter.cuda_stream, mark.sid)
        waiter.wait_event(mark.event)


class TrainStep:
    #: which of the schedule's six roles share a HIP stream (see __init__).  The runtime multiplexes streams onto 4 hardware queues
    #: in creation order, and MORE concurrency is not better (one stream per role on 8 queues: 116.3 ms; on the default 4 queues the
    #: outcome depended on which roles happened to share a queue: 110.6 ms plain, 116.4 once RCCL's own streams had shifted the order).
    #: Three streams beside the caller's, with the sharing chosen: all weight gradients | discriminator A + identity passes |
    #: discriminator B + generator chain A  ->  109.1 ms at batch 8.  A sweep over ~90 partitions of the roles (tools/layout_search.py)
    #: found "012312" 0.4 ms faster at batch 8 but slower at batches 2, 16, 32 and 64 (45.1 vs 40.5 ms at batch 2): this one is the
    #: most even across batch sizes
    #: ... (round 4) with the f16x2 convolutions (65 ms steps, the kernel mix changed) the same sweep puts "012201" -- generators' weight
    #: gradients with the identity passes | discriminators' weight gradients with generator chain B | both discriminator branches --
    #: 3.8 % ahead: 62.4 against 64.7-65.1 ms, two runs each (profiles/r04_layout_search_f16x2.log)
    stream_layout = "012201"
    #: ... the exact-f32 kernels (``precision`` "f32" / "f32_direct": 110 ms steps, another kernel mix) keep round 3's layout: 73.2-73.5 img/s
    #: with it against 71.8-72.1 with "012201"
    stream_layout_f32 = "001212"
    #: ... and with a communicator, whose own streams shift the stream -> hardware-queue assignment: generator chain A on a stream of
    #: its own (111.5 -> 109.4 ms at world 1, three runs each; without a communicator this layout costs 112.9 ms)
    stream_layout_comm = "001232"
    #: (experiments) one HIP stream priority per layout digit, e.g. [0, -1, -1]; None = all default
    stream_priorities = None
    #: data-parallel runs: gradient all-reduces on side streams under the remaining backward work (False: on the main stream, in place)
    overlap_exchange = True
    #: the multi-stream schedule is used from this many pixels per batch on (``overlap_wgrad`` permitting); tests set 0
    overlap_min_pixels = 2 * 256 * 256
    #: the two-chain generator schedule inside a hipGraph capture as well (False: the captured step keeps the single-chain form)
    capture_two_chains = True
    #: eager steps: chain A on the forked stream and chain B on the caller's (round 2's arrangement, which a capture cannot hold):
    #: 109.7 against 111.4 ms per step at batch 8 (same box, two runs each) -- the capturable arrangement puts chain B's backward and the
    #: discriminators' update on streams that share more
    eager_chain_A_forked = True
    #: inside a hipGraph capture: weight gradients on their side stream (True) or in line with the backward chain (False).  Every
    #: weight gradient on the side stream is a fork + join in the graph (~250 per step), which the runtime's graph executor pays for
    #: at replay: 121.3 ms per step against 115.6 in line (batch 8, MI355X; eager streams: 110.7)
    #: "deferred": on the side stream, but the launches of one backward pass are collected and enqueued there in one batch when the
    #: pass is over -- one fork + join per backward pass in the graph (ops.end_wgrad)
    capture_side_wgrad = False
    #: eager multi-stream steps: the discriminators' packed-weight images are written on a branch stream under the generators' forward
    #: (False: behind the generators' images on the caller's stream, as in captured and single-stream steps)
    pack_D_aside = os.environ.get("FAOCTASR_PACK_D_ASIDE", "1") != "0"

    def __init__(self, netG_A2B=None, netG_B2A=None, netD_A=None, netD_B=None, device="cuda", lr=1.3e-4, betas=(0.9, 0.999),
                 beta1=0.25, beta2=10.0, beta3=2.0, beta4=0.5, beta5=0.5, ssim_weight=0.0, whf_weight=0.0, dwt_levels=1,
                 process_group=None, distributed=None, init=True, precision="f32", overlap_wgrad=True,
                 reproducible_forward=False):
        """``precision``: "f32" = exact fp32 MFMA contraction (default); "bf16x3" = the convolutions' three GEMMs on the bf16 matrix
        cores with hi/lo-split operands (16 significant bits: step-0 losses within ~1e-4 of "f32"); "f16x2" = the same kernels on
        fp16 hi/lo-split operands scaled per tensor by a power of two (22 significant bits; per-layer error against fp64 at or below
        the exact-f32 kernels', csrc/split16.h).  Maps narrower than 24 and 1-channel stems / heads run on the f32 kernels in every mode."""
        if precision not in ops.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(ops.PRECISIONS))
        self.precision = precision
        #: forward convolutions without split-K atomics (FAOCTASR_CONV_NO_SPLIT_K): the forward pass, the losses and every activation
        #: mask are then bit-reproducible and two runs differ only by the summation order of the gradient atomics (~2e-6 relative on
        #: the arenas instead of up to 5e-3); ~5 % slower at batch 8 (narrow discriminator maps under-fill the chip)
        self.reproducible_forward = bool(reproducible_forward)
        dev = torch.device(device)
        made = netG_A2B is None
        self.netG_A2B = (netG_A2B or NetworkA2B()).to(dev)            # train.py:73-76
        self.netG_B2A = (netG_B2A or NetworkB2A()).to(dev)
        self.netD_A = (netD_A or FS_DiscriminatorA(1)).to(dev)
        self.netD_B = (netD_B or FS_DiscriminatorB(1)).to(dev)
        if made and init:                                             # train.py:84-88
            for n in (self.netG_A2B, self.netG_B2A, self.netD_A, self.netD_B):
                n.apply(weights_init_normal)
        self.w = dict(beta1=beta1, beta2=beta2, beta3=beta3, beta4=beta4, beta5=beta5)     # train.py:50-54
        self.ssim_weight, self.whf_weight, self.dwt_levels = ssim_weight, whf_weight, dwt_levels
        self.dwt_loss = DWTForward(J=dwt_levels, wave="haar", mode="reflect").to(dev) if whf_weight else None
        # train.py:102-103: one AdamW per side, lr 1.3e-4, betas (0.9, 0.999), default eps/weight_decay
        self.opt_G = ParamArena(live_parameters(self.netG_A2B) + live_parameters(self.netG_B2A), lr, betas)
        self.opt_D = ParamArena(live_parameters(self.netD_A) + live_parameters(self.netD_B), lr, betas)
        self.fake_A_buffer, self.fake_B_buffer = ReplayBuffer(), ReplayBuffer()         # train.py:125-126
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self._targets = {}
        self.device = dev
        self.comm = None
        #: second HIP stream for the weight gradients (ops.wgrad_stream); ``overlap_wgrad=False`` keeps everything on one stream
        self.overlap_wgrad = bool(overlap_wgrad)
        # roles -> streams (``stream_layout``: roles with the same digit share a stream).  Roles: generators' weight gradients,
        # discriminators' weight gradients, discriminator A's branch (frozen pass + update phase), discriminator B's branch, identity
        # passes, generator chain A.
        self._side = self._side_D = self._idt = self._aba = self._branch = None
        if dev.type == "cuda":
            made = {}
            layout = self.stream_layout_comm if self.distributed else (self.stream_layout_f32 if precision in ("f32", "f32_direct") else self.stream_layout)
            prio = self.stream_priorities
            roles = [made.setdefault(ch, torch.cuda.Stream(device=dev, priority=int(prio[int(ch)]) if prio else 0)) for ch in layout]
            self._side, self._side_D, bA, bB, self._idt, self._aba = roles
            self._branch = (bA, bB)
        #: ops.PackPlan per (batch shapes, precision): every packed-weight image a step of that shape uses, repacked in ONE launch at
        #: the step's start.  Per shape, because an image belongs to one (layer, N, H, W): the trailing partial batch of an epoch
        #: (the reference's DataLoader has no drop_last) gets a plan of its own after its first step instead of invalidating the
        #: full batches'.  A step without a plan packs inside its convolution calls; ops._wpack orders those across streams by events.
        self._pack_plans = {}
        if self.distributed:
            if dev.type != "cuda":
                raise KernelError("TrainStep(distributed=True) needs a GPU: the exchange is RCCL behind the C ABI")
            if dev.index is not None:
                torch.cuda.set_device(dev)             # the communicator binds the calling thread's current device
            self.comm = GradComm(process_group)
            self.sync_replicas()

    def sync_replicas(self, root=0):
        """Replicas start identical (what DDP's constructor does): rank ``root``'s parameter arenas, AdamW moments and the
        buffers of the four networks go to every rank."""
        for a in (self.opt_G, self.opt_D):
            for t in (a.flat, a.exp_avg, a.exp_avg_sq):
                self.comm.broadcast(t, root)
        self.sync_buffers(root)

    def sync_buffers(self, root=0):
        """Rank ``root``'s module buffers (BatchNorm running statistics AND the ``num_batches_tracked`` counters) to every rank.
        During training each replica's running statistics follow its own shard (they are not read by a training-mode forward);
        a DDP wrap of the reference would re-broadcast rank 0's before every forward (``broadcast_buffers=True``), so rank 0's
        checkpoint is the same under both, and the other ranks' buffers differ until this is called: call it before an
        evaluation or a checkpoint written by a rank other than ``root``."""
        if self.comm is None:
            return
        for net in (self.netG_A2B, self.netG_B2A, self.netD_A, self.netD_B):
            for m in net.modules():                     # counters the HIP BatchNorm keeps on the host reach the tensor first
                if hasattr(m, "_flush_counter"):
                    m._flush_counter()
            for b in net.buffers():
                if not b.numel():
                    continue
                if b.dtype == torch.float32:
                    self.comm.broadcast(b, root)
                elif b.dtype == torch.int64:            # 8 bytes travel as two floats
                    self.comm.broadcast(b.reshape(-1).view(torch.float32), root)

    def _wgrad_side(self, for_D=False):
        """The stream weight gradients are enqueued on, or None = on the stream of the operation that produced them.  Inside a
        hipGraph capture every cross-stream edge becomes a graph dependency the runtime synchronises at replay (experiment knob
        ``capture_side_wgrad``)."""
        capturing = torch.cuda.is_current_stream_capturing()
        if not self.capture_side_wgrad and capturing:
            return None
        ops.wgrad_defer = capturing and self.capture_side_wgrad == "deferred"
        return self._side_D if for_D else self._side

    # -- pieces of the loop body ---------------------------------------------------------
    def targets(self, B):
        t = self._targets.get(B)
        if t is None:
            t = (torch.ones(B, device=self.device), torch.zeros(B, device=self.device))     # train.py:119-123
            self._targets[B] = t
        return t

    def forward_generators(self, real_A, real_B, critics=None, idt=None):
        """train.py:173-214.  ``critics`` = (stream for netD_A, stream for netD_B): the frozen discriminators' passes of the adversarial
        terms (train.py:221-222) are enqueued on those streams as soon as their fake exists, so that their narrow-map kernels run under
        the remaining generator passes (and, because autograd replays a node on its forward's stream, under the generators' backward).
        ``idt`` = (stream, weight-gradient stream): each identity pass (train.py:177,200) runs on that stream together with its own loss
        term AND that term's backward -- a third of the generators' backward work then runs under their forward, which has no other
        concurrent partner (gradients add linearly and are accumulated with atomics, so the split of ``loss_G.backward()`` into
        three calls changes no value beyond summation order).  A network's BatchNorm running statistics still see its three passes
        in the reference's order: the main stream waits for the identity pass's forward before that network's next pass."""
        G_A2B, G_B2A = self.netG_A2B, self.netG_B2A
        o = {}
        main = torch.cuda.current_stream(self.device) if (critics or idt) else None

        def critic(net, fake, st, key):
            if st is None:
                return
            _wait(st, main)
            fake.record_stream(st)
            with torch.cuda.stream(st):
                o[key] = net(fake)
            o[key].record_stream(main)

        def identity(net, first, second, real, key):
            """``o[key]`` = third output of net(first, second); with ``idt`` also its loss term and that term's backward."""
            if idt is None:
                _, _, o[key] = net(first, second)
                return None
            st, side = idt
            _wait(st, main)
            for t in (first, second, real):
                t.record_stream(st)
            with torch.cuda.stream(st):
                _, _, o[key] = net(first, second)
                ev = _Mark(st)
                term = ops.l1_loss(real, o[key], self.w["beta2"])           # train.py:230-231
                ops.wgrad_stream = side
                try:
                    term.backward()
                finally:
                    ops.end_wgrad()
            o[key].record_stream(main)
            term.record_stream(main)
            o["loss_" + key] = term.detach()
            return ev

        hf, lf = ops.freq_split(real_A, 10, 8)
        ev = identity(G_B2A, hf, lf, real_A, "idt_A")                        # netG_B2A's first pass of the step
        _, hf_feature_A, o["fake_B"] = G_A2B(lf, hf)
        if critics:
            critic(self.netD_B, o["fake_B"], critics[1], "pred_fake_B")
        o["hf_feature_A"] = hf_feature_A.detach()
        hf, lf = ops.freq_split(o["fake_B"], 5, 14)
        _after(main, ev)
        o["hf_feature_recovered_A"], _, o["recovered_A"] = G_B2A(hf, lf)
        hf, lf = ops.freq_split(real_B, 5, 14)
        hf_feature_B, _, o["fake_A"] = G_B2A(hf, lf)
        if critics:
            critic(self.netD_A, o["fake_A"], critics[0], "pred_fake_A")
        ev = identity(G_A2B, lf, hf, real_B, "idt_B")                        # netG_A2B's second pass (after fake_B, before recovered_B)
        o["hf_feature_B"] = hf_feature_B.detach()
        hf, lf = ops.freq_split(o["fake_A"], 10, 8)
        _after(main, ev)
        _, o["hf_feature_recovered_B"], o["recovered_B"] = G_A2B(lf, hf)
        if critics:
            for st in critics:
                _wait(main, st)
        return o

    def _generators_two_chains(self, real_A, real_B, chain_A_forked):
        """The generator phase (train.py:173-236) as two chains on two streams:

            chain A:  fake_B = A2B(real_A), recovered_A = B2A(fake_B), loss_cycle_ABA + loss_GAN_A2B, THEIR BACKWARD
            chain B:  fake_A = B2A(real_B), recovered_B = A2B(fake_A), loss_cycle_BAB + loss_GAN_B2A
            identity passes (stream ``_idt``) with their loss terms and backward; each chain's frozen discriminator on a branch stream

        Chain A's loss terms depend on nothing chain B computes, so its backward -- a third of the generators' backward work -- runs
        under chain B's forward.  Each network's three passes keep the reference's order (A2B: real_A, real_B, fake_A; B2A: real_A,
        fake_B, real_B) through events, so the BatchNorm running statistics are updated in the same sequence; a discriminator's
        frozen pass is ordered before its update phase the same way (its running statistics are plain read-modify-writes).
        Returns (o, L, root): ``root`` is what is left to back-propagate (chain B's terms); every total that mixes streams is formed
        by ``step`` after it has joined them (``_chain_B_terms``).

        ``chain_A_forked`` says which chain sits on the caller's stream; the other one runs on ``_aba``.
          * False -- chain A on the caller's stream: the ONLY arrangement a hipGraph capture can hold.  When a forked stream waits on
            an event of another forked stream, the HIP runtime lists the waiter as a "parallel capture stream" of the other, and
            ``hip::Stream::EndCapture`` walks those lists recursively without a visited set: two forked streams that wait on each
            other recurse until the stack overflows (profiles/r03_capture_crash_gdb_*.log).  Waits to and from the
            capture's ORIGIN stream are never listed.  With chain A -- which both other roles depend on -- on the origin, the waits
            among forked streams form a DAG: ``_idt`` waits only on the caller, ``_aba`` on the caller and on ``_idt``, chain B's
            frozen discriminator runs on chain B's own stream.  ``_CaptureGuard`` enforces the rule.
          * True -- chain A on ``_aba``, chain B on the caller's (eager steps only, ``eager_chain_A_forked``: 109.7 against 111.4 ms
            at batch 8): ``_idt`` then waits on an ``_aba`` event and ``_aba`` on an ``_idt`` event -- harmless outside a capture."""
        G_A2B, G_B2A, w = self.netG_A2B, self.netG_B2A, self.w
        ones, _ = self.targets(real_A.shape[0])
        main = torch.cuda.current_stream(self.device)
        SA, SB = (self._aba, main) if chain_A_forked else (main, self._aba)
        I, side = self._idt, self._wgrad_side()
        cB = self._branch[1]                     # chain A's frozen discriminator (D_B): a branch stream other than chain B's
        if cB.cuda_stream == SB.cuda_stream:
            cB = self._branch[0]
        cA = self._branch[0] if chain_A_forked else SB     # chain B's frozen discriminator (D_A): on chain B's own stream when that is forked
        o, L = {}, {}
        hfA, lfA = ops.freq_split(real_A, 10, 8)            # both filter pairs are (first) used on the caller's stream
        hfB, lfB = ops.freq_split(real_B, 5, 14)
        ev_in = _Mark(main)
        for t in (hfA, lfA, hfB, lfB, real_A, real_B):
            t.record_stream(self._aba)
            t.record_stream(I)

        def identity(net, first, second, real, key, after):
            _after(I, ev_in)
            _after(I, after)
            with torch.cuda.stream(I):
                _, _, o[key] = net(first, second)
                ev = _Mark(I)
                term = ops.l1_loss(real, o[key], w["beta2"])
                ops.wgrad_stream = side
                try:
                    term.backward()
                finally:
                    ops.end_wgrad()
            return ev, term.detach()

        def critic(net, fake, st, src):
            """The frozen discriminator pass of ``fake`` (made on ``src``) on ``st``: (prediction, mark after it)."""
            _wait(st, src)
            fake.record_stream(st)
            with torch.cuda.stream(st):
                pred = net(fake)
                return pred, _Mark(st)

        ev_idt_A, idt_A = identity(G_B2A, hfA, lfA, real_A, "idt_A", None)                  # B2A pass 1
        # ---- chain A, forward
        _after(SA, ev_in)
        with torch.cuda.stream(SA):
            _, hf_feature_A, o["fake_B"] = G_A2B(lfA, hfA)                                  # A2B pass 1
            ev_a2b_1 = _Mark(SA)
            o["hf_feature_A"] = hf_feature_A.detach()
            pred_B, ev_pred_B = critic(self.netD_B, o["fake_B"], cB, SA)
            hf, lf = ops.freq_split(o["fake_B"], 5, 14)
            _after(SA, ev_idt_A)
            o["hf_feature_recovered_A"], _, o["recovered_A"] = G_B2A(hf, lf)                # B2A pass 2
            ev_b2a_2 = _Mark(SA)
        # ---- chain B, forward; enqueued before chain A's backward so that the two overlap
        _after(SB, ev_in)
        _after(SB, ev_b2a_2)
        with torch.cuda.stream(SB):
            hf_feature_B, _, o["fake_A"] = G_B2A(hfB, lfB)                                  # B2A pass 3
            o["hf_feature_B"] = hf_feature_B.detach()
            ev_fake_A = _Mark(SB)
            pred_A, ev_pred_A = critic(self.netD_A, o["fake_A"], cA, SB)
            hf, lf = ops.freq_split(o["fake_A"], 10, 8)
        ev_idt_B, idt_B = identity(G_A2B, lfB, hfB, real_B, "idt_B", ev_a2b_1)              # A2B pass 2 (after pass 1)
        _after(SB, ev_idt_B)
        with torch.cuda.stream(SB):
            _, o["hf_feature_recovered_B"], o["recovered_B"] = G_A2B(lf, hf)                # A2B pass 3
            _after(SB, ev_pred_A)
            pred_A.record_stream(SB)
            L["loss_GAN_B2A"] = ops.mse_loss(pred_A, ones, w["beta5"])
            L["loss_cycle_BAB"] = ops.l1_loss(o["recovered_B"], real_B, w["beta3"]) + \
                ops.bce_with_logits(o["hf_feature_B"], o["hf_feature_recovered_B"], w["beta1"])
            root = L["loss_GAN_B2A"] + L["loss_cycle_BAB"]
            extra_B = self._extension_terms(o["recovered_B"], real_B)       # opt-in SSIM / wavelet-HF terms: one half per chain
            for k, v in extra_B.items():
                root = root + v
        # ---- chain A, losses and backward
        _after(SA, ev_pred_B)
        pred_B.record_stream(SA)
        with torch.cuda.stream(SA):
            L["loss_GAN_A2B"] = ops.mse_loss(pred_B, ones, w["beta4"])
            L["loss_cycle_ABA"] = ops.l1_loss(o["recovered_A"], real_A, w["beta3"]) + ops.bce_with_logits(o["hf_feature_A"], o["hf_feature_recovered_A"])
            chain_A = L["loss_GAN_A2B"] + L["loss_cycle_ABA"]
            extra_A = self._extension_terms(o["recovered_A"], real_A)
            for k, v in extra_A.items():
                chain_A = chain_A + v
            ops.wgrad_stream = side
            try:
                chain_A.backward()
            finally:
                ops.end_wgrad()
        # ---- the caller's stream picks up the fakes (they feed the replay buffers and the discriminator phase) behind BOTH frozen
        # discriminator passes: a discriminator's update phase runs on a branch stream ordered behind the caller only, and its
        # BatchNorm running statistics must see the frozen pass first.  The loss scalars are read after the final joins.
        for ev in (ev_a2b_1, ev_pred_B, ev_fake_A, ev_pred_A):
            _after(main, ev)
        for k in ("loss_GAN_A2B", "loss_cycle_ABA"):
            L[k] = L[k].detach()
        for t in list(o.values()) + list(L.values()) + list(extra_A.values()) + list(extra_B.values()) + [root, pred_A, pred_B, idt_A, idt_B]:
            t.record_stream(main)
        o["pred_fake_A"], o["pred_fake_B"] = pred_A, pred_B
        # totals that mix the chains are formed by ``step`` after it has joined every stream
        self._chain_B_terms = (root, (L["loss_GAN_A2B"], L["loss_cycle_ABA"], idt_A, idt_B),
                               {k: (extra_A[k].detach(), v) for k, v in extra_B.items()})
        return o, L, root

    def _extension_terms(self, rec, real):
        """One image pair's share of the opt-in terms of ``generator_loss`` (SSIM: train.py:234; wavelet-HF L1)."""
        t = {}
        if self.ssim_weight:
            t["loss_ssim"] = self.ssim_weight * (1 - ops.ssim(rec, real))
        if self.whf_weight:
            acc = 0
            _, yh_r = self.dwt_loss(rec)
            _, yh_t = self.dwt_loss(real)
            for a, b in zip(yh_r, yh_t):
                acc = acc + ops.l1_loss(a, b, self.whf_weight)
            t["loss_whf"] = acc
        return t

    def generator_loss(self, o, real_A, real_B):
        """train.py:221-236 (+ the opt-in SSIM term of the commented line train.py:234 and a wavelet-HF L1 term).  ``L["_root"]`` is
        what remains to be back-propagated (everything, unless ``forward_generators`` already did the identity terms)."""
        w = self.w
        ones, _ = self.targets(real_A.shape[0])
        L = {}
        L["loss_GAN_A2B"] = ops.mse_loss(o["pred_fake_B"] if "pred_fake_B" in o else self.netD_B(o["fake_B"]), ones, w["beta4"])
        L["loss_GAN_B2A"] = ops.mse_loss(o["pred_fake_A"] if "pred_fake_A" in o else self.netD_A(o["fake_A"]), ones, w["beta5"])
        L["loss_cycle_ABA"] = ops.l1_loss(o["recovered_A"], real_A, w["beta3"]) + ops.bce_with_logits(o["hf_feature_A"], o["hf_feature_recovered_A"])
        L["loss_cycle_BAB"] = ops.l1_loss(o["recovered_B"], real_B, w["beta3"]) + \
            ops.bce_with_logits(o["hf_feature_B"], o["hf_feature_recovered_B"], w["beta1"])
        total = L["loss_GAN_A2B"] + L["loss_GAN_B2A"] + L["loss_cycle_ABA"] + L["loss_cycle_BAB"]
        done = None
        if "loss_idt_A" in o:
            done = o["loss_idt_A"] + o["loss_idt_B"]
            L["loss_idt"] = done
        else:
            L["loss_idt"] = ops.l1_loss(real_A, o["idt_A"], w["beta2"]) + ops.l1_loss(real_B, o["idt_B"], w["beta2"])
            total = total + L["loss_idt"]
        if self.ssim_weight:
            L["loss_ssim"] = self.ssim_weight * ((1 - ops.ssim(o["recovered_A"], real_A)) + (1 - ops.ssim(o["recovered_B"], real_B)))
            total = total + L["loss_ssim"]
        if self.whf_weight:
            t = 0
            for rec, real in ((o["recovered_A"], real_A), (o["recovered_B"], real_B)):
                _, yh_r = self.dwt_loss(rec)
                _, yh_t = self.dwt_loss(real)
                for a, b in zip(yh_r, yh_t):
                    t = t + ops.l1_loss(a, b, self.whf_weight)
            L["loss_whf"] = t
            total = total + t
        L["_root"] = total
        L["loss_G"] = total if done is None else total.detach() + done
        return L

    def _discriminator_phase(self, L, o, real_A, real_B, _static, branches, side):
        """train.py:242-266 without the optimizer step: both discriminator losses and their backward passes.  ``branches``: one
        stream per discriminator (their updates share nothing), ``side``: the stream for their weight gradients."""
        ones, zeros = self.targets(real_A.shape[0])
        set_requires_grad([self.netD_A, self.netD_B], True)
        self.opt_D.zero_grad()
        fake_A = self.fake_A_buffer.push_and_pop(o["fake_A"]) if _static is None else self.fake_A_buffer.apply(o["fake_A"], *_static["plan_A"])
        fake_B = self.fake_B_buffer.push_and_pop(o["fake_B"]) if _static is None else self.fake_B_buffer.apply(o["fake_B"], *_static["plan_B"])
        main = torch.cuda.current_stream(self.device) if branches[0] is not None else None
        ops.wgrad_stream = side
        try:
            for key, net, real, fake, st in (("loss_D_A", self.netD_A, real_A, fake_A, branches[0]), ("loss_D_B", self.netD_B, real_B, fake_B, branches[1])):
                if st is None:
                    L[key] = ops.mse_loss(net(real), ones, 0.5) + ops.mse_loss(net(fake.detach()), zeros, 0.5)
                    L[key].backward()
                    continue
                _wait(st, main)
                with torch.cuda.stream(st):                      # autograd runs each node's backward on its forward's stream
                    L[key] = ops.mse_loss(net(real), ones, 0.5) + ops.mse_loss(net(fake.detach()), zeros, 0.5)
                    L[key].backward()
                L[key].record_stream(main)
        finally:
            ops.end_wgrad()
        return fake_A, fake_B            # alive until the branches are joined

    def _join_discriminator_phase(self, branches, side):
        _join(torch.cuda.current_stream(self.device), branches)
        ops.join_wgrad_stream(side)

    def step(self, real_A, real_B, sync=False, keep=False, _static=None):
        """One iteration of train.py:166-269.  Returns the losses as device scalars (``sync=True``: floats).
        ``_static`` (GraphedTrainStep): device tensors replacing the host-decided pieces -- replay-buffer index plans and AdamW scalars.

        Stream schedule (``overlap_wgrad``; the arithmetic and its order per tensor are the reference's): weight gradients go to a
        side stream; the discriminator phase reads only the detached fakes and the discriminator weights -- nothing the generators'
        backward produces -- so once every packed-weight image is current it is enqueued on two branch streams BEFORE
        ``loss_G.backward()`` and its narrow-map, low-occupancy kernels run under the generators' large ones."""
        ops.conv_precision = ops.PRECISIONS[self.precision]
        ops.reproducible_forward = self.reproducible_forward
        misses = ops.pack_misses
        plan_key = (tuple(real_A.shape), tuple(real_B.shape), self.precision)
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        streams = self.overlap_wgrad and self._side is not None and (capturing or real_A.numel() >= self.overlap_min_pixels)
        plans = self._pack_plans.get(plan_key)
        # every packed-weight image of a step of this shape in two launches: the generators' on the caller's stream, the discriminators'
        # (two thirds of the bytes: their exact-f32 narrow-map images) on a branch stream in eager multi-stream steps -- nothing reads them
        # before the generators' forward has run, and a reader on another stream waits for the launch's event (ops._wpack)
        packed = False
        if plans is not None:
            aside = self._branch[0] if (streams and not capturing and self.pack_D_aside) else None
            packed = plans[0].run() and plans[1].run(aside)
        if not packed:
            self._pack_plans.pop(plan_key, None)
        # a plan built after this step (a pack miss: new shape, or a weight that moved) then holds exactly the images THIS step used --
        # cleared every step, not only on steps without a plan: otherwise the images of other batch shapes and of evaluation calls
        # touched since the last plan-less step would be swept into the rebuilt plan and repacked on every step
        ops.clear_touched()
        # below ~2 x 256^2 pixels per batch the step is bound by the host's enqueue rate, and the extra events / stream switches
        # of the schedule cost more than the concurrency returns (batch 1 at 256^2: 41.8 vs 39.3 ms; batch 2: 48.3 vs 52.8)
        # ... a CAPTURED step has no host in its way: there the schedule pays at every size (batch 1: 29.9 against 35.4 ms per replay)
        if capturing:
            _guard.begin(torch.cuda.current_stream(self.device).cuda_stream)
        set_requires_grad([self.netD_A, self.netD_B], False)                # train.py:219 (before the first discriminator pass)
        self.opt_G.zero_grad()                                              # train.py:220 (before the first backward of a generator term)
        # (a step without a plan -- the first of its shape -- packs inside its convolution calls; ops._wpack orders a later reader on
        # another stream behind that launch by an event, so the schedule, and with it the order of the two gradient exchanges on the
        # communicator, depends on the batch shape alone, never on rank-local cache state)
        multi = streams
        two_chains = multi and (self.capture_two_chains or not capturing)
        if two_chains:
            o, L, root = self._generators_two_chains(real_A, real_B, chain_A_forked=self.eager_chain_A_forked and not capturing)
        else:
            o = self.forward_generators(real_A, real_B, self._branch if multi else None, (self._idt, self._side) if multi else None)
            # (2) generators, train.py:218-239
            L = self.generator_loss(o, real_A, real_B)
            root = L.pop("_root")
        side_G, side_D = (self._wgrad_side(), self._wgrad_side(True)) if streams else (None, None)
        branches = self._branch if streams else (None, None)
        early_D = streams
        held, d_reduced = None, False
        try:
            if early_D:
                held = self._discriminator_phase(L, o, real_A, real_B, _static, branches, side_D)
                if self.distributed and self.overlap_exchange and side_D is not None:
                    # the discriminators' gradients are complete once their branches and their weight-gradient stream have drained:
                    # their exchange (the larger arena) starts here and runs under the generators' backward
                    for st in branches:
                        _wait(side_D, st)
                    with torch.cuda.stream(side_D):
                        self.opt_D.all_reduce(self.group, self.comm)
                    d_reduced = True
            ops.wgrad_stream = side_G
            try:
                root.backward()
            finally:
                ops.end_wgrad()
                if multi:                        # the identity terms' / chain A's backward ran on their own streams (BatchNorm affine gradients
                    # there), the frozen discriminator passes' input gradients on the branch streams
                    _join(torch.cuda.current_stream(self.device), (self._idt, self._aba) + tuple(self._branch))
                ops.join_wgrad_stream(side_G)
            if two_chains:               # the chains' terms were computed on their streams: the totals are formed after the join
                root_B, terms, halves = self._chain_B_terms
                total = root_B.detach()
                for t in terms:
                    total = total + t
                L["loss_idt"] = terms[2] + terms[3]
                for k, (ha, hb) in halves.items():
                    L[k] = ha + hb.detach()
                    total = total + ha
                L["loss_G"] = total
                self._chain_B_terms = None
            hyper_G = None if _static is None else _static["hyper_G"]
            g_update_aside = self.distributed and streams and self.overlap_exchange and side_G is not None
            if g_update_aside:
                # the generators' gradient exchange and AdamW touch nothing the discriminator phase reads: they run on the side
                # stream under it and are joined before the discriminators' update
                _wait(side_G, torch.cuda.current_stream(self.device))
                if d_reduced:                    # one collective at a time on the communicator, in the same order on every rank
                    _wait(side_G, side_D)
                with torch.cuda.stream(side_G):
                    self.opt_G.all_reduce(self.group, self.comm)
                    self.opt_G.step(1.0 / self.world, hyper_G)
            else:
                if self.distributed:
                    self.opt_G.all_reduce(self.group, self.comm)
                self.opt_G.step(1.0 / self.world, hyper_G)
            # (3) discriminators, train.py:242-269
            if not early_D:
                held = self._discriminator_phase(L, o, real_A, real_B, _static, branches, side_D)
        finally:
            try:
                if streams:
                    self._join_discriminator_phase(branches, side_D)
                    ops.join_wgrad_stream(side_G)
            finally:
                _guard.end()
        del held
        if self.distributed and not d_reduced:
            self.opt_D.all_reduce(self.group, self.comm)
        self.opt_D.step(1.0 / self.world, None if _static is None else _static["hyper_D"])
        if ops.pack_misses != misses and not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # some convolution packed its own weights: first step of this shape / precision -> collect the images this step used
            if len(self._pack_plans) >= 8:
                self._pack_plans.pop(next(iter(self._pack_plans)))
            self._pack_plans[plan_key] = (ops.PackPlan(self.opt_G.params, ops.conv_precision), ops.PackPlan(self.opt_D.params, ops.conv_precision))
        ops.conv_precision = 0
        ops.reproducible_forward = False
        out = {k: v.detach() for k, v in L.items()}
        if sync:
            out = {k: float(v) for k, v in out.items()}
        if keep:
            out["tensors"] = {k: v.detach() for k, v in o.items()}
        return out

    # -- learning-rate schedule (train.py:105-110,287-288) -------------------------------------
    def set_lr(self, lr_G, lr_D=None):
        self.opt_G.lr = float(lr_G)
        self.opt_D.lr = float(lr_G if lr_D is None else lr_D)

    def lr_step(self, base_lr, lr_lambda, epoch):
        """torch.optim.lr_scheduler.LambdaLR semantics: lr = base_lr * lr_lambda(epoch) for both optimizers
        (``lr_lambda`` e.g. ``faoctasr.LambdaLR(n_epochs, offset, decay_epoch).step``)."""
        self.set_lr(base_lr * lr_lambda(epoch))

    def grad_norms(self):
        """Per-network gradient L2 norms (diagnostics / parity tests; host-side reduction)."""
        r = {}
        for name, net in (("A2B", self.netG_A2B), ("B2A", self.netG_B2A), ("D_A", self.netD_A), ("D_B", self.netD_B)):
            sq = 0.0
            for _, p in live_parameters(net):
                if p.grad is not None:
                    sq += float((p.grad.double() ** 2).sum())
            r[name] = sq ** 0.5
        return r


class GraphedTrainStep:
    """The whole train step as ONE captured hipGraph (BASELINE config 5): ~3000 kernel launches per step become one
    graph launch, which removes the host enqueue time (39 ms/step) from small-batch steps.

    What varies from step to step is moved off the host path: the replay buffers become ``DeviceReplayBuffer``s (the reference's
    random decisions are still drawn on the host, in the reference's order, but reach the device as two small index tensors),
    and AdamW reads its learning rate and bias corrections from device memory (``faoctasr_adamw_step_dev``).  Capturing does not
    advance the training state: the warm-up steps that populate the allocator, the packed-weight buffers and the cached
    circulants run on a snapshot that is restored before the capture.  A data-parallel step is captured the same way: the two
    gradient all-reduces are ``faoctasr_grad_allreduce`` calls on the capturing stream, so RCCL's kernels become graph nodes
    (BASELINE config 5); every rank must construct and replay its graph in lockstep, as with any collective."""

    def __init__(self, ts, real_A, real_B, warmup=2):
        self.ts = ts
        dev = ts.device
        B = real_A.shape[0]
        self.real_A, self.real_B = real_A.detach().clone(), real_B.detach().clone()
        ts.fake_A_buffer = DeviceReplayBuffer.adopt(ts.fake_A_buffer)
        ts.fake_B_buffer = DeviceReplayBuffer.adopt(ts.fake_B_buffer)
        for b in (ts.fake_A_buffer, ts.fake_B_buffer):
            b._ensure(self.real_A)
        self._bns = [m for net in (ts.netG_A2B, ts.netG_B2A, ts.netD_A, ts.netD_B) for m in net.modules() if hasattr(m, "_pending_batches")]
        snap = self._snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                ts.step(self.real_A, self.real_B)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._restore(snap)
        mk = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
        self._static = {"plan_A": (mk(B, torch.long), mk(B, torch.long)), "plan_B": (mk(B, torch.long), mk(B, torch.long)),
                        "hyper_G": mk(8, torch.float32), "hyper_D": mk(8, torch.float32)}
        for k in ("plan_A", "plan_B"):                                   # valid indices for the capture pass itself
            self._static[k][0].fill_(ts.fake_A_buffer.max_size + 1)
            self._static[k][1].fill_(ts.fake_A_buffer.max_size)
        pend = [m._pending_batches for m in self._bns]
        ops.invalidate_weight_cache()                                    # every first use inside the graph packs its weights
        ops.reset_scale_arenas()                                         # f16x2: every absmax slot of the graph is zeroed inside the graph
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.losses = ts.step(self.real_A, self.real_B, _static=self._static)
        ops.reset_scale_arenas()                                         # ... and eager code never draws slots from the graph's pool
        self._bn_delta = [m._pending_batches - p for m, p in zip(self._bns, pend)]
        for m, p in zip(self._bns, pend):                                # the capture pass executed nothing
            m._pending_batches = p
        ops.invalidate_weight_cache()

    def _snapshot(self):
        ts = self.ts
        return {"arenas": [(a, a.flat.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone(), a.step_count) for a in (ts.opt_G, ts.opt_D)],
                "buffers": [(b, b.clone()) for net in (ts.netG_A2B, ts.netG_B2A, ts.netD_A, ts.netD_B) for b in net.buffers()],
                "pending": [m._pending_batches for m in self._bns],
                "replay": [(r, r.count, r.store.clone()) for r in (ts.fake_A_buffer, ts.fake_B_buffer)],
                "random": random.getstate()}

    def _restore(self, snap):
        with torch.no_grad():
            for a, flat, m, v, n in snap["arenas"]:
                a.flat.copy_(flat); a.exp_avg.copy_(m); a.exp_avg_sq.copy_(v); a.step_count = n
            for b, val in snap["buffers"]:
                b.copy_(val)
            for r, count, store in snap["replay"]:
                r.count = count
                r.store.copy_(store)
        for m, p in zip(self._bns, snap["pending"]):
            m._pending_batches = p
        random.setstate(snap["random"])
        ops.invalidate_weight_cache()

    def step(self, real_A, real_B, sync=False):
        """One train step = one graph launch.  The returned loss tensors are the graph's static outputs (overwritten by the next step)."""
        ts = self.ts
        B = self.real_A.shape[0]
        self.real_A.copy_(real_A, non_blocking=True)
        self.real_B.copy_(real_B, non_blocking=True)
        for key, buf in (("plan_A", ts.fake_A_buffer), ("plan_B", ts.fake_B_buffer)):      # the reference's draw order: A, then B
            src, dst = buf.plan(B)
            self._static[key][0].copy_(torch.tensor(src, dtype=torch.long))
            self._static[key][1].copy_(torch.tensor(dst, dtype=torch.long))
        for key, opt in (("hyper_G", ts.opt_G), ("hyper_D", ts.opt_D)):
            opt.step_count += 1
            self._static[key].copy_(torch.tensor(opt.hyper_values(opt.step_count, 1.0 / ts.world), dtype=torch.float32))
        self.graph.replay()
        for m, d in zip(self._bns, self._bn_delta):
            m._pending_batches += d
        ops.invalidate_weight_cache()
        return {k: float(v) for k, v in self.losses.items()} if sync else self.losses
