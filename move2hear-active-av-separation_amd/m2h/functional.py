"""Differentiable front ends: torch.autograd.Function wrappers whose forward AND backward arithmetic run in libm2h.so.

PyTorch's autograd engine only orders the calls and accumulates ``.grad`` (plumbing); every gradient value is produced by
a HIP kernel:
  conv / linear   dgrad = forward igemm engine on m2h_pack_dgrad_weight phase matrices; wgrad = m2h_conv_wgrad_f32;
                  bias = m2h_bias_grad; fused ReLU/LeakyReLU = m2h_act_bwd
Replaces torch's Conv2d/Linear autograd in audio_separation/rl/ppo/ppo.py:159-161 (update_pol) and :228-230 (update_sep).
"""
import contextlib
import ctypes
import os
import threading
import weakref

import torch

from . import _lib, ops


_carry_tuning = [False]   # set by carry_tuning(): tests / tools that A/B engines through a backward pass


def carry_tuning(on=True):
    """Make the autograd Functions below carry the forward thread's TUNING KNOBS (ops.debug_set) into their backward, as they always
    carry its arithmetic mode.  Off by default: production code sets no knobs, and the snapshot is a library call per Function."""
    _carry_tuning[0] = bool(on)


def carries_math_mode(cls):
    """Class decorator for the autograd Functions whose backward launches igemm-engine kernels: the forward records the calling
    thread's arithmetic (ops.math_mode(), thread-local) and the backward -- which autograd runs on ITS thread -- computes in it."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._m2h_math = ops.math_mode()
        ctx._m2h_tuning = ops.tuning_snapshot() if _carry_tuning[0] else None
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        with ops.math_scope(ctx._m2h_math):
            if ctx._m2h_tuning is not None:     # the forward thread's tuning knobs (thread-local in the library, include/m2h_tuning.h)
                with ops.tuning_scope(ctx._m2h_tuning):
                    return bwd(ctx, *grads)
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _conv_args(x, x2, n_out, kh, kw, stride, pad, Ho, Wo):
    B, H, W, C0 = x.shape
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), C0, (x2.shape[3] if x2 is not None else 0)
    a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, Ho, Wo
    a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = stride, kh, kw, 1, -pad, 1, -pad
    a.conv_transpose, a.N = 0, n_out
    a.Ho, a.Wo, a.os, a.ph, a.pw, a.ldc, a.out_mode = Ho, Wo, 1, 0, 0, n_out, 0
    return a


_flat_optimizers = weakref.WeakSet()   # FlatAdam instances (m2h/optim.py): their flat gradient buffers offer each weight's gradient a home


def grad_slot(w):
    """Where the gradient of weight `w` will be wanted: if w lives in a FlatAdam's flat parameter buffer, the matching slice of that
    optimizer's flat GRADIENT buffer, shaped like w -- a weight-gradient kernel that writes there spares the optimizer's gather copy
    (FlatAdam._gather skips gradients that already are views of their slice).  Handed out once per zero_grad() and weight (a second
    backward pass before the next zero_grad gets None and its gradient is accumulated by autograd as usual); None for any other tensor."""
    if not w.is_cuda or not w.is_contiguous():
        return None
    ptr = w.data_ptr()
    for opt in list(_flat_optimizers):
        if not getattr(opt, "_built", False) or opt.flat_p.device != w.device:
            continue
        base = opt.flat_p.data_ptr()
        if base <= ptr < base + 4 * opt.n:
            off = (ptr - base) // 4
            if off + w.numel() > opt.n or off in opt._slots_used:
                return None
            opt._slots_used.add(off)
            return opt.flat_g[off:off + w.numel()].view(w.shape)
    return None


# ----------------------------------------------------------------------------------------------------------------
# Weight gradients as a side branch of a captured step.  In a backward pass the input-gradient kernels form the dependent chain
# (layer L's dgrad feeds layer L-1's); each layer's WEIGHT gradient hangs off that chain and nothing reads it before the optimizer.
# Enqueued on one stream, the 10-60 us weight-gradient launches of these batch sizes sit between the chain's links.
#
# What this stack allows (ROCm 7.2; profiles/r06_wgrad_side_per_layer_ab.txt, tools/r06_fork_probe.py): every cross-stream edge of a
# replayed graph costs ~19 us (one fork PER LAYER made the passive step 3.15 instead of 2.40 ms), and a side stream of a side stream may
# only be joined by the stream the capture began on (joined by the intermediate stream: hipStreamEndCapture crashes).  Hence the form:
# inside ``wgrad_side_branches()``, while a HIP graph is being captured, a layer whose gradient has a home in its optimizer's flat buffer
# (grad_slot) does not launch its weight gradient, it DEFERS it (operands kept alive by the closure); ``wgrad_flush_point(x)`` -- an
# identity in the forward, placed by a model where its backward is half done (the U-Net's bottleneck) -- launches what was deferred so
# far on ONE side stream behind ONE event while the chain goes on; what is deferred after the last flush point is launched on the
# chain's own stream by ``flush_deferred_wgrads()`` (the trainer calls it after backward()).  The optimizer's reads
# (``join_wgrad_branches``: FlatAdam.captured_step / _gather) wait for the side streams.  Same kernels, same values.
# Used by the passive training step (2.41 -> 2.37 ms, profiles/r06_wgrad_side_ab.txt).  In update_pol's epoch the same idea -- the
# recurrent encoder's two weight gradients started on the third encoder's branch instead of the chain ahead of the fork -- returned
# nothing (35.7 / 36.2 against 35.5 / 35.0 ms per cycle, same file) and is not built in.
# ----------------------------------------------------------------------------------------------------------------
_wgrad_side = {"on": False}
_wgrad_deferred = {}    # (device index, stream id of the backward) -> [(launch, operands, arithmetic mode, tuning knobs)]
_wgrad_pending = {}     # (device index, side stream id) -> (side stream holding weight-gradient launches nobody waited for yet, id of the stream they forked from)


@contextlib.contextmanager
def wgrad_side_branches(enabled=True):
    """Inside the block (and inside a graph capture) weight-gradient launches of this module's Functions are deferred to the next
    flush (see above); the caller flushes every stream a backward ran on, on that stream.  The flag is process-wide, not thread-local:
    autograd runs the backward on its own thread."""
    from . import graphs
    prev = _wgrad_side["on"]
    _wgrad_side["on"] = bool(enabled) and graphs.parallel_branches and os.environ.get("M2H_WGRAD_SIDE", "1") != "0"    # (the variable: A/B runs)
    try:
        yield
        if _wgrad_deferred:
            raise RuntimeError("m2h.wgrad_side_branches: weight gradients were deferred and never launched (call flush_deferred_wgrads() after backward())")
    finally:
        _wgrad_side["on"] = prev
        _wgrad_deferred.clear()
        if not prev:
            _wgrad_pending.clear()    # (a body that raised may leave side streams nobody joined: a later capture must not wait for them)


def _wgrad_launch(dev, reads, fn, slot):
    """fn() -- one layer's weight-gradient launches, writing to `slot` -- now, or deferred to the next flush (see above: switch on, a
    capture in progress, and the gradient has a home that exists before the launch)."""
    if slot is None or not _wgrad_side["on"] or dev.type != "cuda" or ops.timing_enabled() or not torch.cuda.is_current_stream_capturing():
        return fn()
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    _wgrad_deferred.setdefault(key, []).append((fn, reads, ops.math_mode(), ops.tuning_snapshot() if _carry_tuning[0] else None))
    # A second tensor object over the same memory: autograd's AccumulateGrad keeps the tensor it is handed as .grad only when nobody else
    # holds it (otherwise it CLONES it -- here: a copy of memory the deferred launch has not written yet); the closure holds `slot`.
    return slot.detach()


def flush_deferred_wgrads(device=None, side=False):
    """Launches the weight gradients deferred by the backward that runs on the current stream: on that stream (side False), or on its
    weight-gradient side stream behind one event (side True; the operands are handed to the side stream for the allocator --
    ``record_stream``: inside a capture their memory is then not re-used before the capture ends)."""
    cur = torch.cuda.current_stream(device)
    items = _wgrad_deferred.pop((cur.device.index, cur.cuda_stream), None)
    if not items:
        return

    def run():
        for fn, _reads, mode, knobs in items:
            with ops.math_scope(mode):
                if knobs is not None:
                    with ops.tuning_scope(knobs):
                        fn()
                else:
                    fn()

    if not side:
        return run()
    from . import graphs
    st = graphs.side_stream(cur.device, ("wgrad", cur.cuda_stream))
    ev = torch.cuda.Event()
    ev.record(cur)
    st.wait_event(ev)
    with torch.cuda.stream(st):
        run()
    for _fn, reads, _m, _k in items:
        for t in reads:
            if t is not None and t.numel():
                t.record_stream(st)
    _wgrad_pending[(cur.device.index, st.cuda_stream)] = (st, cur.cuda_stream)


class _WgradFlushPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        flush_deferred_wgrads(g.device, side=True)
        return g


def wgrad_flush_point(x):
    """Identity.  In a backward pass captured inside ``wgrad_side_branches()``, the point where the weight gradients deferred so far are
    launched on the side stream (nothing otherwise: no node is added when the switch is off)."""
    if not _wgrad_side["on"] or not x.requires_grad:
        return x
    return _WgradFlushPoint.apply(x)


def join_wgrad_branches(device=None, forked_from=None):
    """The current stream waits for the weight-gradient launches forked since the last join: those forked from stream `forked_from`, or
    (None) all of the device's -- a step that reads the whole flat gradient reads every layer's.  The caller must be on the stream the
    capture began on whenever the launches forked from a branch (see above; the passive step does its second network's Adam there)."""
    if _wgrad_deferred:
        raise RuntimeError("m2h.join_wgrad_branches: weight gradients are still deferred (flush_deferred_wgrads() comes first)")
    if not _wgrad_pending:
        return
    cur = torch.cuda.current_stream(device)
    want = forked_from.cuda_stream if forked_from is not None else None
    for key in [k for k, (_s, org) in _wgrad_pending.items() if k[0] == cur.device.index and (want is None or org == want)]:
        cur.wait_stream(_wgrad_pending.pop(key)[0])


def conv_wgrad(x, x2, dy, n_out, kh, kw, stride, pad, gate=None, gate_slope=1.0, torch_ci=None, out=None):
    """Packed weight gradient [n_out, kh*kw*(C0+C1)] of a conv whose NHWC inputs were x (+x2) and NHWC output grad is dy.
    gate: the layer's forward output y; dy is then read as dy * (y > 0 ? 1 : gate_slope) (m2h_conv_wgrad_gated_f32: image-row shapes only).
    torch_ci: return nn.Conv2d's own layout [n_out, torch_ci, kh, kw] instead (m2h_conv_wgrad_torch_f32: no permute copy afterwards).
    out: write the gradient there (grad_slot(w): the weight's place in its optimizer's flat gradient buffer)."""
    B, Ho, Wo, N = dy.shape
    a = _conv_args(x, x2, n_out, kh, kw, stride, pad, Ho, Wo)
    lib = _lib.load()
    K = kh * kw * (x.shape[3] + (x2.shape[3] if x2 is not None else 0))
    shape = (n_out, K) if torch_ci is None else (n_out, int(torch_ci), kh, kw)
    if out is not None and (tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32):
        raise RuntimeError("m2h.conv_wgrad: out must be a contiguous fp32 tensor of shape %s" % (shape,))
    dw = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        nbytes = lib.m2h_conv_wgrad_workspace_bytes(ctypes.byref(a))
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
        M = B * Ho * Wo
        meta = {"kernel": "wgrad_f32", "M": M, "N": n_out, "K": K, "flops": 2.0 * M * n_out * K, "bytes": 4.0 * (x.numel() + dy.numel() + dw.numel())}
        if torch_ci is not None:
            ops._timed("conv_wgrad", meta, x.device,
                       lambda: _lib.check(lib.m2h_conv_wgrad_torch_f32(ctypes.byref(a), ops._ptr(dy), N, ops._ptr(gate) if gate is not None else None,
                                                                       float(gate_slope), ops._ptr(dw), int(torch_ci), ops._stream(x)), "m2h_conv_wgrad_torch_f32"))
        elif gate is not None:
            ops._timed("conv_wgrad", meta, x.device,
                       lambda: _lib.check(lib.m2h_conv_wgrad_gated_f32(ctypes.byref(a), ops._ptr(dy), N, ops._ptr(gate), float(gate_slope), ops._ptr(dw),
                                                                       ops._stream(x)), "m2h_conv_wgrad_gated_f32"))
        else:
            ops._timed("conv_wgrad", meta, x.device,
                       lambda: _lib.check(lib.m2h_conv_wgrad_f32(ctypes.byref(a), ops._ptr(dy), N, ops._ptr(dw), ops._stream(x)), "m2h_conv_wgrad_f32"))
    return dw


def conv_wgrad_dgrad_fused_supported(x, n_out=32):
    """True when m2h_conv_wgrad_dgrad_fused_f32 takes a 3x3 / 1 / 1 conv over x [B, H, 32, 32] -> n_out in the calling thread's arithmetic."""
    B, H, W, _C = x.shape
    a = _conv_args(x, None, n_out, 3, 3, 1, 1, H, W)
    return bool(_lib.load().m2h_conv_wgrad_dgrad_fused_supported(ctypes.byref(a)))


def conv_wgrad_dgrad_fused(x, dy2, wp_next, gate, gate_slope, torch_ci, out=None):
    """Weight gradient [32, torch_ci, 3, 3] of a 3x3 / 1 / 1 conv (+ ReLU: gate = its forward output) whose output gradient is the INPUT gradient
    of the next 3x3 conv, made inside the kernel from dy2 [B, H, W, 16] and that conv's packed weight wp_next [16, 288]
    (m2h_conv_wgrad_dgrad_fused_f32: the 32-channel gradient tensor is never stored).  bf16x3 arithmetic, image-row shapes only."""
    B, H, W, C = x.shape
    a = _conv_args(x, None, 32, 3, 3, 1, 1, H, W)
    lib = _lib.load()
    shape = (32, int(torch_ci), 3, 3)
    if out is not None and (tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32):
        raise RuntimeError("m2h.conv_wgrad_dgrad_fused: out must be a contiguous fp32 tensor of shape %s" % (shape,))
    if tuple(dy2.shape) != (B, H, W, 16) or not dy2.is_contiguous() or wp_next.numel() != 16 * 9 * C or tuple(gate.shape) != (B, H, W, 32):
        raise RuntimeError("m2h.conv_wgrad_dgrad_fused: dy2 must be [B, H, W, 16] contiguous, wp_next [16, 288], gate [B, H, W, 32]")
    dw = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        nbytes = lib.m2h_conv_wgrad_workspace_bytes(ctypes.byref(a))
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
        _lib.check(lib.m2h_conv_wgrad_dgrad_fused_f32(ctypes.byref(a), ops._ptr(dy2), ops._ptr(wp_next), ops._ptr(gate), float(gate_slope), ops._ptr(dw),
                                                      int(torch_ci), ops._stream(x)), "m2h_conv_wgrad_dgrad_fused_f32")
    return dw


def pack_dgrad_weight(w4d, stride, pad):
    Co, Ci, KH, KW = w4d.shape
    wp = torch.empty((stride * stride, Ci, (KH // stride) * (KW // stride) * Co), device=w4d.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(w4d.device):
        _lib.check(lib.m2h_pack_dgrad_weight(ops._ptr(w4d), ops._ptr(wp), Co, Ci, KH, KW, stride, pad, ops._stream(w4d)), "m2h_pack_dgrad_weight")
    return wp


def conv_dgrad(dy, w4d, in_hw, stride, pad, ci_out=None, wp=None):
    """Input gradient [B,H,W,Ci] of Conv2d(w4d [Co,Ci,KH,KW], stride, pad) given the NHWC output gradient dy [B,Ho,Wo,Co].
    wp: the weight's phase matrices (m2h_pack_dgrad_weight) when the caller keeps them packed."""
    B, Ho, Wo, Co = dy.shape
    _, Ci, KH, KW = w4d.shape
    H, W = in_hw
    s = stride
    if wp is None:
        wp = pack_dgrad_weight(w4d, s, pad)
    dx = torch.empty((B, H, W, Ci), device=dy.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(dy.device):
        for ph in range(s):
            for pw in range(s):
                Hq, Wq = (H - ph + s - 1) // s, (W - pw + s - 1) // s
                if Hq <= 0 or Wq <= 0:
                    continue
                a = _lib.ConvArgs()
                a.src0, a.src1, a.C0, a.C1 = dy.data_ptr(), None, Co, 0
                a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, Ho, Wo, Hq, Wq
                a.stride, a.nth, a.ntw = 1, KH // s, KW // s
                a.mulh, a.offh = -1, (ph + pad - (ph + pad) % s) // s
                a.mulw, a.offw = -1, (pw + pad - (pw + pad) % s) // s
                a.conv_transpose, a.wp, a.N = 0, wp[ph * s + pw].data_ptr(), Ci
                a.scale, a.shift, a.slope, a.cls_table, a.cls_val = None, None, 1.0, None, None
                a.dst, a.Ho, a.Wo, a.os, a.ph, a.pw, a.ldc, a.out_mode = dx.data_ptr(), H, W, s, ph, pw, Ci, 0
                nbytes = lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a))
                ws = torch.empty((nbytes + 3) // 4, device=dy.device, dtype=torch.float32) if nbytes else None
                a.workspace, a.workspace_bytes = (ws.data_ptr() if ws is not None else None), nbytes
                M = B * Hq * Wq
                K = (KH // s) * (KW // s) * Co
                meta = {"kernel": ops.igemm_config(Ci), "M": M, "N": Ci, "K": K, "flops": 2.0 * M * Ci * K}
                ops._timed("conv_dgrad", meta, dy.device,
                           lambda: _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), ops._stream(dy)), "m2h_conv_igemm_f32(dgrad)"))
    return dx


def act_bwd(dy, y, slope):
    out = torch.empty_like(dy)
    lib = _lib.load()
    with torch.cuda.device(dy.device):
        _lib.check(lib.m2h_act_bwd(ops._ptr(dy), ops._ptr(y), float(slope), ops._ptr(out), dy.numel(), ops._stream(dy)), "m2h_act_bwd")
    return out


def act_bwd_bias(dy, y, slope):
    """(act_bwd(dy, y, slope), bias_grad of it) from one pass over dy (m2h_act_bwd_bias); dy, y: [..., N] contiguous."""
    N = dy.shape[-1]
    M = dy.numel() // N
    out = torch.empty_like(dy)
    db = torch.empty(N, device=dy.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(dy.device):
        ws = torch.empty((lib.m2h_bias_grad_workspace_bytes(M, N) + 3) // 4, device=dy.device, dtype=torch.float32)
        _lib.check(lib.m2h_act_bwd_bias(ops._ptr(dy), ops._ptr(y), float(slope), ops._ptr(out), ops._ptr(db), M, N, ops._ptr(ws), ops._stream(dy)),
                   "m2h_act_bwd_bias")
    return out, db


def bias_grad(dy2d):
    M, N = dy2d.shape
    db = torch.empty(N, device=dy2d.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(dy2d.device):
        ws = torch.empty((lib.m2h_bias_grad_workspace_bytes(M, N) + 3) // 4, device=dy2d.device, dtype=torch.float32)
        _lib.check(lib.m2h_bias_grad(ops._ptr(dy2d), ops._ptr(db), M, N, ops._ptr(ws), ops._stream(dy2d)), "m2h_bias_grad")
    return db


_param_epoch = 0


def bump_param_epoch():
    """Called by FlatAdam.step: its kernels update parameters through raw pointers, which torch's version counters do not
    see, so packed-weight memos also key on this epoch."""
    global _param_epoch
    _param_epoch += 1


def param_epoch():
    return _param_epoch


_pack_memos = weakref.WeakSet()


class _PackMemo:
    """Packed operands of ONE weight tensor: the forward pack (Conv2d [Co][KH*KW*ci_pad] / ConvTranspose2d [4][Co][4*Cin]) and,
    once a backward pass has asked for it, the input-gradient pack.  Each is re-packed only when the weights may have changed
    (data_ptr, version counter, optimizer epoch) and then IN PLACE, so a packed buffer keeps its address for as long as the
    weight keeps its shape (a captured HIP graph holds that address).  refresh_pack_memos() brings every memo up to date with
    ONE batched launch (m2h_pack_batch) -- before a graph replay, and at the top of a training step so that the step's own
    forward / backward calls all hit."""

    def __init__(self):
        self.fwd = [None, None, None]   # [key, packed tensor, spec]
        self.bwd = [None, None, None]
        self.src = None                 # (Parameter, view shape or None): how to find the weight again at refresh time
        _pack_memos.add(self)

    def __deepcopy__(self, memo):
        """A copied module gets an empty memo of its own, registered like any other (copy.deepcopy would clone the packed buffers
        and skip __init__, so refresh_pack_memos would never see the copy and a replayed graph would read stale packs)."""
        return _PackMemo()

    # compatibility with callers that look at the forward pack directly
    @property
    def key(self):
        return self.fwd[0]

    @property
    def val(self):
        return self.fwd[1]

    @staticmethod
    def _wkey(w):
        return (w.data_ptr(), w._version, _param_epoch)

    @staticmethod
    def _plan(w, spec):
        """(shape of the packed tensor, m2h_pack_batch kind, p[6]) for a weight and a pack spec."""
        if spec[0] == "conv":            # Conv2d forward: [Co][Ci][KH][KW] -> [Co][KH*KW*ci_pad]
            Co, Ci, KH, KW = w.shape
            return (Co, KH * KW * spec[1]), _lib.PACK_CONV, (Co, Ci, KH, KW, Ci, spec[1])
        if spec[0] == "convT":           # ConvTranspose2d forward: [Cin][Co][4][4] -> [4][Co][4*Cin]
            Cin, Co = w.shape[0], w.shape[1]
            return (4, Co, 4 * Cin), _lib.PACK_CONVT, (Cin, Co, 0, 0, 0, 0)
        if spec[0] == "dgrad":           # Conv2d input gradient: stride^2 phase matrices
            Co, Ci, KH, KW = w.shape
            st, pad = spec[1], spec[2]
            return (st * st, Ci, (KH // st) * (KW // st) * Co), _lib.PACK_DGRAD, (Co, Ci, KH, KW, st, pad)
        if spec[0] == "fc_dgrad":        # full-spatial conv (a Linear): dX = dY @ Wp as one GEMM, Wp^T [KH*KW*c_in][Co]
            Co, Ci, KH, KW = w.shape
            return (KH * KW * spec[1], Co), _lib.PACK_FC_DGRAD, (Co, Ci, KH, KW, Ci, spec[1])
        if spec[0] == "dgrad_as_convT":  # Conv2d(4,2,1) input gradient = conv_transpose2d(dy, w): w [Co][Ci][4][4] IS a ConvTranspose2d weight [in=Co][out=Ci]
            Co, Ci = w.shape[0], w.shape[1]
            return (4, Ci, 4 * Co), _lib.PACK_CONVT, (Co, Ci, 0, 0, 0, 0)
        if spec[0] == "convT_dgrad":     # ConvTranspose2d input gradient = a Conv2d with the weight read as [out=Cin][in=Co]
            Cin, Co = w.shape[0], w.shape[1]
            return (Cin, 16 * Co), _lib.PACK_CONV, (Cin, Co, 4, 4, Co, Co)
        raise ValueError(spec)

    def _item(self, slot, w, spec):
        """Pack item for `slot` (self.fwd / self.bwd) if it is stale for weight w, else None; marks the slot fresh."""
        key = self._wkey(w) + tuple(spec)
        if slot[0] == key:
            return None
        shape, kind, prm = self._plan(w, spec)
        wd = w.detach()
        if not wd.is_contiguous():
            wd = wd.contiguous()
        if slot[1] is None or tuple(slot[1].shape) != shape or slot[1].device != wd.device:
            slot[1] = torch.empty(shape, device=wd.device, dtype=torch.float32)
        slot[0], slot[2] = key, tuple(spec)
        self.src = (w._base, tuple(w.shape)) if w._base is not None else (w, None)
        return (kind, wd, slot[1], prm)

    def _get(self, slot, w, spec):
        item = self._item(slot, w, spec)
        if item is not None:
            ops.pack_batch([item])
        return slot[1]

    def get(self, w, ci_pad):
        return self._get(self.fwd, w, ("conv", ci_pad))

    def get_convT(self, w):
        return self._get(self.fwd, w, ("convT",))

    def get_bwd(self, w, spec):
        return self._get(self.bwd, w, spec)

    def mark_fresh(self):
        """Host bookkeeping only: the packed buffers ARE up to date with the current weights (a replayed HIP graph packed them)."""
        self.stale_items()

    def stale_items(self, force=False):
        if self.src is None:
            return []
        if force:
            for slot in (self.fwd, self.bwd):
                slot[0] = None
        base, shape = self.src
        w = base if shape is None else base.view(shape)
        out = []
        for slot in (self.fwd, self.bwd):
            if slot[2] is not None:
                item = self._item(slot, w, slot[2])
                if item is not None:
                    out.append(item)
        return out


_refresh_hooks = weakref.WeakSet()   # objects with a sync() that brings derived weight tensors up to date before the packs (FusedAudioPair)


def memos_of(*modules):
    """The packed-weight memos held by `modules` and their sub-modules (attributes that are a _PackMemo or a list / tuple of them)."""
    out = []

    def walk(v):
        if isinstance(v, _PackMemo):
            out.append(v)
        elif isinstance(v, (list, tuple)):
            for m in v:
                walk(m)      # (nested lists too: RNNStateEncoder keeps a pair of memos per layer)

    for root in modules:
        for mod in root.modules():
            for v in vars(mod).values():
                walk(v)
    return out


def refresh_pack_memos(hooks=True, only=None, force=False):
    """Re-packs (in place) every packed operand whose source weights changed since it was packed, with one batched launch per
    48 tensors.  Called before a HIP-graph replay (the graph reads the packed buffers by address and contains no pack kernels)
    and at the top of a captured training step.
    hooks: also bring derived weight tensors (FusedAudioPair's block-diagonal copies) up to date first.  Skipped inside a
    HIP-graph capture (a capture of some OTHER model must not trip over them: they are rebuilt outside captures, and their user
    checks freshness itself) and by callers whose graph does not read them (the update_pol epoch).
    only: restrict the refresh to these memos (``memos_of(network)``): a step that packs each network's weights on that network's own
    stream (the passive training step's two graph branches).
    force: re-pack every slot of the memos whatever their keys say (a graph capture that must contain the pack launches)."""
    if hooks and not torch.cuda.is_current_stream_capturing():
        for h in list(_refresh_hooks):
            h.sync()
    items = []
    for m in (list(_pack_memos) if only is None else only):   # only: the memos of one network (memos_of), the others are the caller's business
        items += m.stale_items(force)
    if items:
        ops.pack_batch(items)


@carries_math_mode
class Conv2dNHWC(torch.autograd.Function):
    """y = act(conv2d(cat(x, x2), w) + b) over NHWC activations; w, b in torch layout ([Co,Ci,KH,KW], [Co]).
    ``deslice``: the output is stored de-sliced in the reference's BHWC layout (memory_nets.py:62-67)."""

    @staticmethod
    def forward(ctx, x, x2, w, b, stride, pad, slope, deslice, memo, name):
        Co, Ci, KH, KW = w.shape
        c_in = x.shape[3] + (x2.shape[3] if x2 is not None else 0)
        wp = memo.get(w, c_in) if memo is not None else ops.pack_conv_weight_ex(w.detach().contiguous(), Ci, c_in)
        y = ops.conv2d_nhwc(x, wp, Co, KH, KW, stride=stride, pad=pad, bias=b.detach() if b is not None else None, slope=slope, x2=x2,
                            deslice=deslice, name=name)
        ctx.cfg = (stride, pad, slope, deslice, Ci, KH, KW, Co)
        ctx.memo = memo
        ctx.has_x2 = x2 is not None
        ctx.save_for_backward(x, x2 if x2 is not None else x.new_empty(0), w, y if slope != 1.0 else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, slope, deslice, Ci, KH, KW, Co = ctx.cfg
        x, x2, w, y = ctx.saved_tensors
        x2 = x2 if ctx.has_x2 else None
        dy = dy.contiguous()
        if deslice:
            # BHWC [B,16*Ho,Wo,Co/16] gradient -> NHWC [B,Ho,Wo,Co]: the same 16-way slice map as the forward input glue
            dy = ops.slice_concat_input(dy, op=0)
            if slope != 1.0:
                raise NotImplementedError("m2h Conv2dNHWC: activation + de-sliced output has no backward")
        need_x = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        # AcousticMem's first conv over the update batch (3x3 / 1 / 1, 32 channels, 32-pixel rows, no bias, no input gradient): the
        # weight gradient is the only reader of dy, and its image-row kernel applies the activation's derivative as it loads dy
        # (the predicate mirrors EVERY condition of the library's image-row rule, csrc/conv_bwd.hip `row3x3`: 3x3 / 1 / 1, one 32-channel source,
        # 32-pixel rows, output grid == image (Ho x Wo == H x W: "direct"), N <= 32 and N % 4 == 0, dy rows N floats apart (contiguous NHWC: ldy % 4),
        # one weight tile (N <= 32, K = 288 <= one k-tile group) -- a shape that passed here and failed there would raise inside backward)
        gated = (slope != 1.0 and ctx.needs_input_grad[2] and not need_x and not ctx.needs_input_grad[3] and x2 is None and
                 (KH, KW, stride, pad) == (3, 3, 1, 1) and x.shape[3] == 32 and x.shape[2] == 32 and Co <= 32 and Co % 4 == 0 and
                 tuple(dy.shape[1:3]) == tuple(x.shape[1:3]) and dy.shape[3] == Co and dy.is_contiguous() and y.is_contiguous() and
                 x.shape[0] * x.shape[1] >= 512 and not ops.timing_enabled())
        gb = None
        if slope != 1.0 and not gated:
            if ctx.needs_input_grad[3] and dy.is_contiguous() and y.is_contiguous():
                dy, gb = act_bwd_bias(dy, y, slope)      # the activation's backward and the bias gradient from one pass over dy
            else:
                dy = act_bwd(dy, y, slope)
        B, Ho, Wo, _ = dy.shape
        gx = gx2 = gw = None
        if ctx.needs_input_grad[2]:
            # the gradient arrives in the weight's own layout [Co, Ci, KH, KW] (split sum + re-layout in one launch)
            slot = grad_slot(w)             # (None unless a FlatAdam owns w: then the kernel writes where the optimizer reads)
            if gated and ops.tuning_snapshot()[21] < 0:   # the image-row weight-gradient kernel is switched off (m2h_tuning_set(21, -1)): two passes
                dy = act_bwd(dy, y, slope)
                gated = False
            if gated:                       # (any failure of the gated launch is a real error: it is not swallowed)
                gw = _wgrad_launch(x.device, (x, dy, y), lambda: conv_wgrad(x, x2, dy, Co, KH, KW, stride, pad, gate=y, gate_slope=slope, torch_ci=Ci, out=slot), slot)
            if gw is None:
                dyw = dy     # (the name `dy` is not rebound below, but a deferred launch must not depend on that)
                gw = _wgrad_launch(x.device, (x, x2, dyw), lambda: conv_wgrad(x, x2, dyw, Co, KH, KW, stride, pad, torch_ci=Ci, out=slot), slot)
        if ctx.needs_input_grad[3] and gb is None:
            gb = bias_grad(dy.view(B * Ho * Wo, Co))
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            if x2 is not None:
                raise NotImplementedError("m2h Conv2dNHWC: input gradient of a two-source conv is not built yet")
            if KH == x.shape[1] and KW == x.shape[2] and pad == 0 and Ho == 1 and Wo == 1:
                # full-spatial conv (a Linear over the NCHW-flattened map, visual_cnn.py:140-141): dX = dY @ Wp as ONE GEMM.
                # The generic phase formulation would walk KH*KW taps per input pixel with a single valid one (144x the work).
                c_in = x.shape[3]
                if ctx.memo is not None:
                    wt = ctx.memo.get_bwd(w, ("fc_dgrad", c_in))                         # [(h,w,c)][Co]
                else:
                    wp = ops.pack_conv_weight_ex(w.detach().contiguous(), Ci, c_in)      # [Co][(h,w,c)]
                    wt = pack_dgrad_weight(wp.view(Co, KH * KW * c_in, 1, 1), 1, 0).view(KH * KW * c_in, Co)
                gx = ops.linear(dy.view(B, Co), wt, None, name="fc.dgrad").view(B, KH, KW, c_in)
            elif (KH, KW, stride, pad) == (4, 4, 2, 1) and x.shape[1] == 2 * Ho and x.shape[2] == 2 * Wo and Ci % 4 == 0 and Ci == x.shape[3]:
                # the U-Net's encoder convs (separator_cnn.py:5-13): dX = conv_transpose2d(dY, W) -- the four sub-pixel phases in ONE
                # launch of the transposed-conv forward (the generic path below is one launch per phase: 4 x 10-19 us per layer)
                wpt = ctx.memo.get_bwd(w, ("dgrad_as_convT",)) if ctx.memo is not None else ops.pack_convT_weight(w.detach().contiguous())
                gx = ops.unet_up_fwd_raw(dy, None, wpt, Ci)
            else:
                wpd = ctx.memo.get_bwd(w, ("dgrad", stride, pad)) if ctx.memo is not None else None
                gx = conv_dgrad(dy, w.detach().contiguous(), (x.shape[1], x.shape[2]), stride, pad, wp=wpd)
            if gx.shape[3] != x.shape[3]:  # channel-padded input (VisualCNN 3 -> 4): padded channels carry no gradient
                gx = torch.nn.functional.pad(gx, (0, x.shape[3] - gx.shape[3]))
        return gx, gx2, gw, gb, None, None, None, None, None, None


def conv2d(x, w, b=None, stride=1, pad=0, slope=1.0, x2=None, deslice=False, memo=None, name="conv2d"):
    return Conv2dNHWC.apply(x, x2, w, b, stride, pad, slope, deslice, memo, name)


def linear(x, w, b=None, slope=1.0, memo=None, name="linear"):
    """y = act(x W^T + b) with autograd (x [M,K], w [N,K])."""
    M, K = x.shape
    y = conv2d(x.reshape(M, 1, 1, K), w.reshape(w.shape[0], K, 1, 1), b, 1, 0, slope, memo=memo, name=name)
    return y.reshape(M, w.shape[0])


# ----------------------------------------------------------------------------------------------------------------
# GRU (sequence with per-step hidden resets), policy heads, losses
# ----------------------------------------------------------------------------------------------------------------
def _lin_nograd(x, w_nk, bias=None, name="linear"):
    return ops.linear(x, w_nk, bias, name=name)


@carries_math_mode
class GRUSequence(torch.autograd.Function):
    """h_t = GRUCell(x_t, h_{t-1} * mask_t) for t < T over N rows; x [T*N, I], h0 [N, H], masks [T*N] -> (out [T*N, H], hT).
    Covers single_forward (T = 1) and seq_forward (rnn_state_encoder.py:74-137).  Backward = BPTT with HIP kernels:
    per step one fused gate-gradient kernel + one recurrent GEMM; the four weight/bias gradients and dx are batched GEMMs."""

    @staticmethod
    def forward(ctx, x, h0, masks, w_ih, w_hh, b_ih, b_hh, T, memos=None):
        """memos: optional (_PackMemo of W_ih, _PackMemo of W_hh): the transposed copies the backward's two input-gradient products read are
        then kept by the memos and refreshed with every other pack (functional.refresh_pack_memos: one batched launch ahead of an update
        epoch) instead of being made inside the backward, on its serial chain."""
        ctx.memos = memos
        N, H = h0.shape
        gi = _lin_nograd(x.contiguous(), w_ih.detach(), b_ih.detach(), "gru.ih")  # [T*N, 3H]
        masks = masks.reshape(T * N).contiguous()
        out = torch.empty((T * N, H), device=x.device, dtype=torch.float32)
        gh = torch.empty((T * N, 3 * H), device=x.device, dtype=torch.float32)
        h = h0.contiguous()
        fused = N <= ops.GRU_STEP_MAX_ROWS and H % 16 == 0  # rollout width: the whole step is one launch
        whh, bhh = w_hh.detach().contiguous(), b_hh.detach()
        for t in range(T):
            sl = slice(t * N, (t + 1) * N)
            if fused:
                h, _ = ops.gru_step(gi[sl], whh, bhh, h, masks[sl], gh_out=gh[sl], out=out[sl])
                continue
            ght = ops.linear(h, whh, None, name="gru.hh", out=gh[sl])      # written in place into the saved buffer
            h = ops.gru_gates(gi[sl], ght, bhh, h, masks[sl], out=out[sl])
        ctx.T = T
        ctx.save_for_backward(x, h0, masks, w_ih, w_hh, b_hh, gi, gh, out)
        ctx.set_materialize_grads(False)      # (the update never differentiates the final state: no zero tensor, no add of it in the backward)
        return out, h

    @staticmethod
    def backward(ctx, g_out, g_hT):
        x, h0, masks, w_ih, w_hh, b_hh, gi, gh, out = ctx.saved_tensors
        T = ctx.T
        N, H = h0.shape
        dev = x.device
        lib = _lib.load()
        g_out = g_out.contiguous() if g_out is not None else torch.zeros_like(out)
        dgi = torch.empty_like(gi)
        dpre = torch.empty_like(gh)
        hpm = torch.empty_like(out)
        dhp = torch.empty((N, H), device=dev)
        # W_hh^T as an [H][3H] "Linear" weight for the recurrent dgrad GEMM
        if ctx.memos is not None:
            whh_t = ctx.memos[1].get_bwd(w_hh.view(3 * H, H, 1, 1), ("dgrad", 1, 0)).view(H, 3 * H)
        else:
            whh_t = pack_dgrad_weight(w_hh.detach().reshape(3 * H, H, 1, 1).contiguous(), 1, 0).view(H, 3 * H)
        fused = N <= ops.GRU_STEP_MAX_ROWS and H % 16 == 0
        dh = g_out[(T - 1) * N:T * N]               # (read only below: no copy)
        if g_hT is not None:
            dh = dh + g_hT  # tiny [N,H] add; hT is rarely used downstream
        dh = dh.contiguous()
        with torch.cuda.device(dev):
            h0c = h0.contiguous()
            for t in range(T - 1, -1, -1):
                sl = slice(t * N, (t + 1) * N)
                hprev = out[(t - 1) * N:t * N] if t > 0 else h0c
                if not fused or t == T - 1:   # (at the rollout width every later step's gate backward rides on the step after it)
                    _lib.check(lib.m2h_gru_gates_bwd(ops._ptr(gi[sl]), ops._ptr(gh[sl]), ops._ptr(b_hh), ops._ptr(hprev),
                                                     ops._ptr(masks[sl]), ops._ptr(dh), ops._ptr(dgi[sl]), ops._ptr(dpre[sl]), ops._ptr(dhp),
                                                     ops._ptr(hpm[sl]), N, H, ops._stream(x)), "m2h_gru_gates_bwd")
                nxt = torch.empty((N, H), device=dev)
                a = g_out[(t - 1) * N:t * N] if t > 0 else None
                if fused and t > 0:
                    # recurrent product + combine of step t, then the gate backward of step t - 1 on the dh just produced: ONE launch
                    sp = slice((t - 1) * N, t * N)
                    hpp = out[(t - 2) * N:(t - 1) * N] if t > 1 else h0c
                    _lib.check(lib.m2h_gru_bwd_step(ops._ptr(dpre[sl]), ops._ptr(whh_t), ops._ptr(a), ops._ptr(dhp), ops._ptr(masks[sl]),
                                                    ops._ptr(nxt), ops._ptr(gi[sp]), ops._ptr(gh[sp]), ops._ptr(b_hh), ops._ptr(hpp),
                                                    ops._ptr(masks[sp]), ops._ptr(dgi[sp]), ops._ptr(dpre[sp]), ops._ptr(hpm[sp]), N, H,
                                                    ops._stream(x)), "m2h_gru_bwd_step")
                elif fused:  # rollout width: recurrent product + combine in one launch
                    _lib.check(lib.m2h_gru_bwd_rec(ops._ptr(dpre[sl]), ops._ptr(whh_t), ops._ptr(a), ops._ptr(dhp), ops._ptr(masks[sl]),
                                                   ops._ptr(nxt), N, H, ops._stream(x)), "m2h_gru_bwd_rec")
                else:
                    rec = _lin_nograd(dpre[sl], whh_t, None, "gru.hh.dgrad")  # dpre @ W_hh  [N,H]
                    _lib.check(lib.m2h_gru_bwd_combine(ops._ptr(a), ops._ptr(rec), ops._ptr(dhp), ops._ptr(masks[sl]), ops._ptr(nxt), N, H,
                                                       ops._stream(x)), "m2h_gru_bwd_combine")
                dh = nxt
        g_h0 = dh if ctx.needs_input_grad[1] else None
        # batched parameter / input gradients
        M = T * N
        I = x.shape[1]
        # (a 1x1 "conv": the packed gradient [3H][K] IS the weight's layout -- written straight into the optimizer's flat gradient buffer)
        g_wih = g_whh = None
        if ctx.needs_input_grad[3]:
            s_ih = grad_slot(w_ih)
            g_wih = _wgrad_launch(dev, (x, dgi), lambda: conv_wgrad(x.reshape(M, 1, 1, I), None, dgi.view(M, 1, 1, 3 * H), 3 * H, 1, 1, 1, 0, out=s_ih), s_ih)
        if ctx.needs_input_grad[4]:
            s_hh = grad_slot(w_hh)
            g_whh = _wgrad_launch(dev, (hpm, dpre), lambda: conv_wgrad(hpm.view(M, 1, 1, H), None, dpre.view(M, 1, 1, 3 * H), 3 * H, 1, 1, 1, 0, out=s_hh), s_hh)
        g_bih = bias_grad(dgi) if ctx.needs_input_grad[5] else None
        g_bhh = bias_grad(dpre) if ctx.needs_input_grad[6] else None
        g_x = None
        if ctx.needs_input_grad[0]:
            if ctx.memos is not None:
                wih_t = ctx.memos[0].get_bwd(w_ih.view(3 * H, I, 1, 1), ("dgrad", 1, 0)).view(I, 3 * H)
            else:
                wih_t = pack_dgrad_weight(w_ih.detach().reshape(3 * H, I, 1, 1).contiguous(), 1, 0).view(I, 3 * H)
            g_x = _lin_nograd(dgi, wih_t, None, "gru.ih.dgrad")
        return g_x, g_h0, None, g_wih, g_whh, g_bih, g_bhh, None, None


class LSTMCell(torch.autograd.Function):
    """(h', c') = LSTM cell pointwise part from the two products gi, gh [N, 4H], the carried cell state and the step's reset mask
    (rnn_state_encoder.py:10-34, 63-69): one launch forward (m2h_lstm_cell), one backward (m2h_lstm_cell_bwd) -- in place of the ~12
    pointwise torch kernels per step of the unfused gate math."""

    @staticmethod
    def forward(ctx, gi, gh, c_prev, mask):
        gi, gh, c_prev = gi.contiguous(), gh.contiguous(), c_prev.contiguous()
        mask = mask.reshape(-1).contiguous()
        N, H = c_prev.shape
        h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
        need = any(ctx.needs_input_grad[:3])
        gates = torch.empty_like(gi) if need else None
        with torch.cuda.device(gi.device):
            _lib.check(_lib.load().m2h_lstm_cell(ops._ptr(gi), ops._ptr(gh), ops._ptr(c_prev), ops._ptr(mask), ops._ptr(h), ops._ptr(c), ops._ptr(gates),
                                                 N, H, ops._stream(gi)), "m2h_lstm_cell")
        if need:
            ctx.save_for_backward(gates, c_prev, mask, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, mask, c = ctx.saved_tensors
        N, H = c_prev.shape
        dpre, dcp = torch.empty_like(gates), torch.empty_like(c_prev)
        dh = dh.contiguous() if dh is not None else None
        dc = dc.contiguous() if dc is not None else None
        with torch.cuda.device(gates.device):
            _lib.check(_lib.load().m2h_lstm_cell_bwd(ops._ptr(dh), ops._ptr(dc), ops._ptr(gates), ops._ptr(c_prev), ops._ptr(mask), ops._ptr(c), ops._ptr(dpre),
                                                     ops._ptr(dcp), N, H, ops._stream(gates)), "m2h_lstm_cell_bwd")
        return dpre, dpre, dcp, None


class PolicyHeads(torch.autograd.Function):
    """(value [M,1], logp_act [M,1], entropy [M], probs [M,A], logp_all [M,A]) from feats and the two Linear heads
    (common/utils.py:42-50, rl/ppo/policy.py:15-23); backward for value / logp_act / entropy."""

    @staticmethod
    def forward(ctx, feats, wa, ba, wc, bc, actions):
        value, logp_all, probs, ent, logp_act = ops.policy_heads(feats.contiguous(), wa.detach(), ba.detach(), wc.detach(), bc.detach(),
                                                                 actions)
        ctx.save_for_backward(feats, wa, wc, logp_all, probs, actions if actions is not None else feats.new_empty(0))
        ctx.has_actions = actions is not None
        ctx.mark_non_differentiable(probs, logp_all)
        ctx.set_materialize_grads(False)      # (no zero tensors -- a fill launch each -- for the outputs nobody differentiates: backward takes None)
        if logp_act is None:
            logp_act = feats.new_zeros((feats.shape[0], 1))
        return value, logp_act, ent, probs, logp_all

    @staticmethod
    def backward(ctx, g_value, g_logp, g_ent, _gp, _gl):
        feats, wa, wc, logp_all, probs, actions = ctx.saved_tensors
        M, H = feats.shape
        A = wa.shape[0]
        ZS = (A + 1 + 3) // 4 * 4
        dev = feats.device
        ge = g_ent.reshape(-1).contiguous() if g_ent is not None else None
        dz = torch.empty((M, ZS), device=dev)
        dfeats = torch.empty((M, H), device=dev)
        gv = g_value.reshape(-1).contiguous() if g_value is not None else None
        gl = g_logp.reshape(-1).contiguous() if (g_logp is not None and ctx.has_actions) else None
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.m2h_policy_heads_bwd(ops._ptr(logp_all), ops._ptr(probs), ops._ptr(actions if ctx.has_actions else None),
                                                ops._ptr(gv), ops._ptr(gl), ops._ptr(ge), ops._ptr(wa.detach()), ops._ptr(wc.detach()), ops._ptr(dz),
                                                ops._ptr(dfeats), M, H, A, ZS, ops._stream(feats)), "m2h_policy_heads_bwd")
        if H % 64 == 0 and ZS in (4, 8) and feats.is_contiguous():
            # both heads' weight and bias gradients in one small launch (the tiled weight-gradient engine + its reduce + the bias launch were 28 us here)
            dw = torch.empty((ZS, H), device=dev)
            db = torch.empty((ZS,), device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.m2h_policy_heads_wgrad(ops._ptr(feats.detach()), ops._ptr(dz), ops._ptr(dw), ops._ptr(db), M, H, ZS, ops._stream(feats)),
                           "m2h_policy_heads_wgrad")
        else:
            dw = conv_wgrad(feats.detach().reshape(M, 1, 1, H), None, dz.view(M, 1, 1, ZS), ZS, 1, 1, 1, 0)  # [ZS, H]
            db = bias_grad(dz)
        return dfeats, dw[:A].contiguous(), db[:A].contiguous(), dw[A:A + 1].contiguous(), db[A:A + 1].contiguous(), None


class PPOLoss(torch.autograd.Function):
    """total = value_loss*value_loss_coef + action_loss - entropy.mean()*entropy_coef (ppo.py:125-154), one kernel computing the
    losses and their gradients; returns (total, stats[4] = (value_loss, action_loss, entropy, total))."""

    @staticmethod
    def forward(ctx, values, logp, entropy, old_values, returns, adv, old_logp, clip, value_loss_coef, entropy_coef, clipped):
        out, gv, gl = ops.ppo_loss(values.contiguous(), logp.contiguous(), old_values.contiguous(), returns.contiguous(), adv.contiguous(),
                                   old_logp.contiguous(), clip, value_loss_coef, clipped, True, entropy.contiguous(), entropy_coef)
        ctx.save_for_backward(gv, gl)
        ctx.ecoef = entropy_coef
        ctx.n = values.numel()
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[3], out

    @staticmethod
    def backward(ctx, g_total, _g_stats):
        gv, gl = ctx.saved_tensors
        key = (gv.device, ctx.n, float(ctx.ecoef))
        ge = _entropy_grads.get(key)               # d total / d entropy[i] = -entropy_coef / n: a constant tensor, made once
        if ge is None:
            ge = torch.full((ctx.n,), -ctx.ecoef / ctx.n, device=gv.device)
            if not torch.cuda.is_current_stream_capturing():      # (a tensor born inside a capture holds nothing until a replay: not cached)
                if len(_entropy_grads) > 16:
                    _entropy_grads.clear()
                _entropy_grads[key] = ge
        u = _unit_grads.get(g_total.device)
        if u is not None and g_total.data_ptr() == u.data_ptr():
            return gv, gl, ge, None, None, None, None, None, None, None, None     # root gradient = unit_grad: no multiplies by one (4 launches per epoch)
        # any other root gradient: a scalar multiply on [n,1] tensors
        return gv * g_total, gl * g_total, ge * g_total, None, None, None, None, None, None, None, None


class L1Loss(torch.autograd.Function):
    """F.l1_loss(pred, gt_comps[..., off:off+1]) with the ground truth read strided in place (ppo.py:212-213)."""

    @staticmethod
    def forward(ctx, pred, gt_comps, off):
        pred = pred.contiguous()
        n = pred.numel()
        loss = torch.empty(1, device=pred.device)
        grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        scratch = torch.empty(1024, device=pred.device)
        lib = _lib.load()
        with torch.cuda.device(pred.device):
            _lib.check(lib.m2h_l1_loss(ops._ptr(pred), ops._ptr(gt_comps), gt_comps.shape[-1], off, ops._ptr(loss), ops._ptr(grad),
                                       ops._ptr(scratch), n, ops._stream(pred)), "m2h_l1_loss")
        if grad is not None:
            ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        u = _unit_grads.get(g.device)
        if u is not None and g.data_ptr() == u.data_ptr():
            return grad, None, None     # d(loss)/d(loss) = the library's own 1.0 (unit_grad): the product is the factor itself, bit for bit
        return grad * g, None, None


@carries_math_mode
class ConvL1NHWC16(torch.autograd.Function):
    """loss = F.l1_loss(deslice(conv3x3(h, w)), gt_comps[..., off]) -- AcousticMem's last conv, its de-slice and update_sep's loss
    (memory_nets.py:16, :62-67; ppo.py:206-216) -- without the de-sliced tensor: the conv stores NHWC, m2h_l1_loss_nhwc16 reads that once and
    leaves d loss / d y in NHWC, where the conv's weight- and input-gradient launches read it (no de-sliced store, no re-slice of the
    gradient: 160 of an update_sep epoch's 840 us were those two round trips)."""

    @staticmethod
    def forward(ctx, h, w, gt_comps, off, memo):
        Co, Ci, KH, KW = w.shape
        if (Co, KH, KW) != (16, 3, 3) or h.shape[1] != 32:
            raise RuntimeError("m2h.conv_l1_nhwc16: a 3x3 conv to 16 bands over 32-row images expected")
        wp = memo.get(w, h.shape[3]) if memo is not None else ops.pack_conv_weight_ex(w.detach().contiguous(), Ci, h.shape[3])
        if gt_comps.shape[-1] == 1 and off == 0 and ops.conv3x3_l1_supported(h) and not ops.timing_enabled():
            # the target already is a plane of its own (update_sep hands it over so, ppo.py) and the shape is the image-row kernels':
            # conv + loss + the loss's gradient in one launch, the conv's output never stored
            loss, dy = ops.conv3x3_l1_nhwc16(h, wp, gt_comps)
        else:
            y = ops.conv2d_nhwc(h, wp, Co, 3, 3, stride=1, pad=1, slope=1.0, name="acoustic_mem.conv1")
            loss, dy = ops.l1_loss_nhwc16(y, gt_comps, off, want_grad=True)
        ctx.save_for_backward(h, w, dy)
        ctx.memo = memo
        return loss

    @staticmethod
    def backward(ctx, g):
        h, w, dy = ctx.saved_tensors
        u = _unit_grads.get(g.device)
        if u is None or g.data_ptr() != u.data_ptr():
            dy = dy * g
        gh = gw = None
        Co, Ci, _, _ = w.shape
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(h, None, dy, Co, 3, 3, 1, 1, torch_ci=Ci, out=grad_slot(w))
        if ctx.needs_input_grad[0]:
            wpd = ctx.memo.get_bwd(w, ("dgrad", 1, 1)) if ctx.memo is not None else None
            gh = conv_dgrad(dy, w.detach().contiguous(), (h.shape[1], h.shape[2]), 1, 1, wp=wpd)
            if gh.shape[3] != h.shape[3]:
                gh = torch.nn.functional.pad(gh, (0, h.shape[3] - gh.shape[3]))
        return gh, gw, None, None, None


def conv_l1_nhwc16(h, w, gt_comps, off=0, memo=None):
    return ConvL1NHWC16.apply(h, w, gt_comps.contiguous(), off, memo)


@carries_math_mode
class AcousticMemL1(torch.autograd.Function):
    """loss = F.l1_loss(deslice(conv1(ReLU(conv0(x)))), gt) with gradients for the two weights only -- update_sep's whole differentiable path
    (rl/ppo/ppo.py:206-226 through rl/models/memory_nets.py:11-16,62-67) as three forward-side and three backward-side launches at the
    image-row shapes in bf16x3 arithmetic: conv0 + ReLU | conv1 + loss + the loss's gradient (m2h_conv3x3_l1_nhwc16: conv1's output never stored) ||
    conv1's weight gradient | conv0's weight gradient with conv1's INPUT gradient and the ReLU gate made inside it
    (m2h_conv_wgrad_dgrad_fused_f32: the 32-channel gradient of h never stored).  The per-layer Functions (Conv2dNHWC, ConvL1NHWC16) stay the
    route of every other shape / arithmetic and compute the same values to fp32 summation order."""

    @staticmethod
    def forward(ctx, x, w0, w1, gt_plane, memo0, memo1):
        wp0 = memo0.get(w0, x.shape[3]) if memo0 is not None else ops.pack_conv_weight_ex(w0.detach().contiguous(), w0.shape[1], x.shape[3])
        wp1 = memo1.get(w1, 32) if memo1 is not None else ops.pack_conv_weight_ex(w1.detach().contiguous(), w1.shape[1], 32)
        h = ops.conv2d_nhwc(x, wp0, 32, 3, 3, stride=1, pad=1, slope=0.0, name="acoustic_mem.conv0")
        loss, dy = ops.conv3x3_l1_nhwc16(h, wp1, gt_plane)
        ctx.save_for_backward(x, h, dy, w0, w1, wp1)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, h, dy, w0, w1, wp1 = ctx.saved_tensors
        u = _unit_grads.get(g.device)
        if u is None or g.data_ptr() != u.data_ptr():
            dy = dy * g
        gw1 = conv_wgrad(h, None, dy, 16, 3, 3, 1, 1, torch_ci=w1.shape[1], out=grad_slot(w1)) if ctx.needs_input_grad[2] else None
        gw0 = None
        if ctx.needs_input_grad[1]:
            gw0 = conv_wgrad_dgrad_fused(x, dy, wp1, h, 0.0, w0.shape[1], out=grad_slot(w0))
        return None, gw0, gw1, None, None, None


def acoustic_mem_l1_supported(x):
    """The shapes / arithmetic AcousticMemL1 runs in: both fused launches available (bf16x3, 32 x 32 x 32 images, B >= 64)."""
    return (x.dim() == 4 and x.shape[3] == 32 and ops.math_mode() == ops.MATH_BF16X3 and not ops.timing_enabled() and ops.conv3x3_l1_supported(x)
            and conv_wgrad_dgrad_fused_supported(x))


def acoustic_mem_l1(x, w0, w1, gt_plane, memo0=None, memo1=None):
    return AcousticMemL1.apply(x, w0, w1, gt_plane.contiguous(), memo0, memo1)


_unit_grads = {}
_entropy_grads = {}


def unit_grad(device):
    """A cached scalar 1.0 on `device` to pass as the root gradient (``loss.backward(unit_grad(dev))``): L1Loss.backward recognises
    it by address and hands its saved gradient on without the elementwise multiply by one (a 110 MB read-modify-write per
    update_sep epoch); any other root gradient takes the general path."""
    device = torch.device(device)
    t = _unit_grads.get(device)
    if t is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a tensor born inside a capture lives in the graph's pool and holds nothing until a replay (and its fill is a graph
            # node): every capture of this package goes through graphs.capture, which creates it first
            raise RuntimeError("m2h.unit_grad: first use inside a graph capture; call functional.unit_grad(device) before capturing")
        t = _unit_grads[device] = torch.ones((), device=device)
    return t


def l1_loss(pred, gt_comps, off=0):
    if pred.numel() != gt_comps.numel() // gt_comps.shape[-1]:
        raise RuntimeError("m2h.l1_loss: pred %s vs gt_comps %s" % (tuple(pred.shape), tuple(gt_comps.shape)))
    return L1Loss.apply(pred, gt_comps.contiguous(), off)


# ----------------------------------------------------------------------------------------------------------------
# passive pre-training pieces: train-mode BN + activation, transposed conv with gradients, binaural L1
# ----------------------------------------------------------------------------------------------------------------
@carries_math_mode
class BNAct(torch.autograd.Function):
    """y = act(BatchNorm2d_train(z)) on NHWC z; running statistics updated in place (momentum 0.1, unbiased variance).
    (fp32 in every arithmetic mode; the decorator is here for the tuning knobs it carries into the backward under carry_tuning(): knob 37.)"""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, slope):
        z = z.contiguous()
        C = z.shape[-1]
        M = z.numel() // C
        dev = z.device
        y = torch.empty_like(z)
        mean = torch.empty(C, device=dev)
        invstd = torch.empty(C, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            ws = torch.empty((lib.m2h_bn_workspace_bytes(M, C) + 3) // 4, device=dev)
            _lib.check(lib.m2h_bn_train_fwd(ops._ptr(z), ops._ptr(gamma.detach()), ops._ptr(beta.detach()), float(eps), float(momentum), float(slope),
                                            ops._ptr(running_mean), ops._ptr(running_var), ops._ptr(mean), ops._ptr(invstd), ops._ptr(y), M, C,
                                            ops._ptr(ws), ops._stream(z)), "m2h_bn_train_fwd")
        ctx.slope = slope
        ctx.save_for_backward(z, y, mean, invstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, y, mean, invstd, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        C = z.shape[-1]
        M = z.numel() // C
        dev = z.device
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, device=dev)
        dbeta = torch.empty(C, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            ws = torch.empty((lib.m2h_bn_workspace_bytes(M, C) + 3) // 4, device=dev)
            _lib.check(lib.m2h_bn_train_bwd(ops._ptr(dy), ops._ptr(y), ops._ptr(z), ops._ptr(mean), ops._ptr(invstd), ops._ptr(gamma.detach()),
                                            float(ctx.slope), ops._ptr(dgamma), ops._ptr(dbeta), ops._ptr(dz), M, C, ops._ptr(ws),
                                            ops._stream(z)), "m2h_bn_train_bwd")
        return dz, dgamma, dbeta, None, None, None, None, None


def bn_act_train(z, bn, slope):
    """Train-mode nn.BatchNorm2d `bn` (parameter container) + activation on NHWC z."""
    y = BNAct.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum if bn.momentum is not None else 0.1, slope)
    sink = getattr(_bn_counters, "sink", None)
    if sink is not None:
        sink.append(bn.num_batches_tracked)
    else:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    return y


_bn_counters = threading.local()


@contextlib.contextmanager
def batched_bn_counters():
    """Inside the block, train-mode BatchNorm calls of this thread collect their ``num_batches_tracked`` counters instead of
    incrementing them one launch each (20 per passive training step); on exit they are all incremented by ONE multi-tensor
    launch.  Same values afterwards (nn.BatchNorm2d's own ``+= 1`` per forward, torch/nn/modules/batchnorm.py)."""
    prev = getattr(_bn_counters, "sink", None)
    _bn_counters.sink = sink = []
    try:
        yield
    finally:
        _bn_counters.sink = prev
        if sink:
            with torch.no_grad():
                torch._foreach_add_(sink, 1)


def _convT_phase_args(x, x2, Co, ph, pw):
    B, H, W, C0 = x.shape
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), C0, (x2.shape[3] if x2 is not None else 0)
    a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, H, W
    a.stride, a.nth, a.ntw = 1, 2, 2
    a.mulh, a.offh, a.mulw, a.offw = 2 * ph - 1, 0, 2 * pw - 1, 0
    a.conv_transpose, a.N = 0, Co
    a.Ho, a.Wo, a.os, a.ph, a.pw, a.ldc, a.out_mode = 2 * H, 2 * W, 2, ph, pw, Co, 0
    return a


@carries_math_mode
class ConvTranspose2dNHWC(torch.autograd.Function):
    """z = conv_transpose2d(cat(x, x2), w, stride 2, pad 1, 4x4) on NHWC (no bias / activation): forward = 4 sub-pixel phase
    GEMMs in one launch; backward: dgrad = an ordinary 4x4/s2/p1 conv of dz with w read as a conv weight (one launch per
    source), wgrad = the wgrad kernel over all four phases in one launch, its ordered reduce scattering to the torch layout."""

    @staticmethod
    def forward(ctx, x, x2, w, memo):
        Cin, Co = w.shape[0], w.shape[1]
        wp = memo.get_convT(w) if memo is not None else ops.pack_convT_weight(w.detach().contiguous())
        z = ops.unet_up_fwd_raw(x, x2, wp, Co)
        ctx.memo = memo
        ctx.has_x2 = x2 is not None
        ctx.save_for_backward(x, x2 if x2 is not None else x.new_empty(0), w)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, x2, w = ctx.saved_tensors
        x2 = x2 if ctx.has_x2 else None
        dz = dz.contiguous()
        Cin, Co = w.shape[0], w.shape[1]
        C0 = x.shape[3]
        B, H, W, _ = x.shape
        gx = gx2 = gw = None
        if ctx.needs_input_grad[2]:
            lib = _lib.load()
            slot = grad_slot(w)

            def wgrad():
                with torch.cuda.device(x.device):
                    a = _convT_phase_args(x, x2, Co, 0, 0)   # the phase geometry; the launch walks all four phases
                    nbytes = lib.m2h_convT_wgrad_workspace_bytes(ctypes.byref(a))
                    ws = torch.empty((nbytes + 3) // 4, device=x.device)
                    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
                    M = B * H * W
                    out = slot if slot is not None else torch.empty_like(w)
                    meta = {"kernel": "wgrad_f32", "M": 4 * M, "N": Co, "K": 4 * Cin, "flops": 2.0 * 4 * M * Co * 4 * Cin}
                    ops._timed("convT_wgrad", meta, x.device,
                               lambda: _lib.check(lib.m2h_convT_wgrad_f32(ctypes.byref(a), ops._ptr(dz), Co, ops._ptr(out), ops._stream(x)),
                                                  "m2h_convT_wgrad_f32"))
                return out

            gw = _wgrad_launch(x.device, (x, x2, dz), wgrad, slot)
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            # [Cin][4][4][Co]: w read as a Conv2d weight [out=Cin][in=Co]
            wconv = ctx.memo.get_bwd(w, ("convT_dgrad",)) if ctx.memo is not None else ops.pack_conv_weight(w.detach().contiguous())
            if ctx.needs_input_grad[0]:
                gx = ops.conv2d_nhwc(dz, wconv[:C0].contiguous(), C0, 4, 4, stride=2, pad=1, slope=1.0, name="convT.dgrad")
            if x2 is not None and ctx.needs_input_grad[1]:
                gx2 = ops.conv2d_nhwc(dz, wconv[C0:].contiguous(), Cin - C0, 4, 4, stride=2, pad=1, slope=1.0, name="convT.dgrad")
        return gx, gx2, gw, None


def conv_transpose2d(x, w, x2=None, memo=None):
    return ConvTranspose2dNHWC.apply(x, x2, w, memo)


class BinL1Loss(torch.autograd.Function):
    """F.l1_loss(masks * (exp(mix) - 1), gt_bin_mag) with gradient w.r.t. masks (passive_trainer.py:270-272)."""

    @staticmethod
    def forward(ctx, masks, mix, gt, cstep):
        loss, grad = ops.bin_l1_loss(mix.contiguous(), masks.contiguous(), gt.contiguous(), cstep, want_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        u = _unit_grads.get(g.device)
        if u is not None and g.data_ptr() == u.data_ptr():
            return grad, None, None, None   # root gradient = unit_grad: the saved gradient itself, no multiply by one
        return grad * g, None, None, None


def bin_l1_loss(masks, mix, gt, cstep=1):
    return BinL1Loss.apply(masks, mix, gt, cstep)
