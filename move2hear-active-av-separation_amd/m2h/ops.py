"""torch-tensor front end of the C-ABI (include/m2h.h).  PyTorch supplies device memory and the current
HIP stream only; all arithmetic happens in libm2h.so.  Every function raises RuntimeError when the library
is missing or a tensor is not a contiguous fp32 CUDA(HIP) tensor -- there is no fallback.
"""
import ctypes
import threading

import torch

from . import _lib

OUT_NHWC = 0
OUT_DESLICE = 1

# Optional per-launch timing (bench.py): when set to a list, every kernel launch is bracketed by two HIP
# events recorded on the launch stream and (name, meta, ev0, ev1) is appended.  None = no overhead.
_timing = None


def set_timing(sink):
    global _timing
    _timing = sink


def timing_enabled():
    """True while bench.py brackets every launch with HIP events (the per-op path is then used so that each kernel is timed)."""
    return _timing is not None


def igemm_config(N):
    """Name of the igemm_f32_kernel width conv_igemm.hip picks for N output channels (skinny-M launches use 32/64-row tiles)."""
    return "igemm_f32<128,128>" if N > 64 else ("igemm_f32<128,64>" if N > 32 else ("igemm_f32<128,32>" if N > 16 else "igemm_f32<128,16>"))


def last_kernel():
    """Label of the calling thread's most recent launch through libm2h (m2h_last_kernel)."""
    return _lib.load().m2h_last_kernel().decode()


def unet_stage_kernels():
    """Labels of the 11 stages of the calling thread's most recent m2h_unet_fwd call (slice, 5 encoder stages, 4 decoder stages,
    last stage + head): the kernels the library's dispatch really launched."""
    lib = _lib.load()
    return [lib.m2h_unet_fwd_stage_kernel(i).decode() for i in range(11)]


def _timed(name, meta, dev, fn):
    if _timing is None:
        return fn()
    st = torch.cuda.current_stream(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r = fn()
    e1.record(st)
    if meta is not None:
        meta = dict(meta, label=last_kernel())   # what the library's dispatch really launched (m2h_last_kernel)
    _timing.append((name, meta, e0, e1))
    return r


def _workspace(nbytes, dev):
    """Split-K scratch from the caching allocator (None when the launch does not split)."""
    if nbytes == 0:
        return None, 0
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    return ws, nbytes


def debug_set(knob, value):
    """Tuning / test knob of the CALLING THREAD (m2h_tuning_set, include/m2h_tuning.h): which of several kernels computing the same values
    the library's dispatch takes.  Thread-local like the arithmetic mode; autograd Functions carry a forward's knobs into their backward
    (functional.carries_math_mode)."""
    _lib.check(_lib.load().m2h_tuning_set(int(knob), int(value)), "m2h_tuning_set")


def tuning_snapshot():
    """The calling thread's tuning knobs as a ctypes int array (m2h_tuning_snapshot)."""
    arr = (ctypes.c_int * _lib.TUNING_KNOBS)()
    _lib.check(_lib.load().m2h_tuning_snapshot(arr, _lib.TUNING_KNOBS), "m2h_tuning_snapshot")
    return arr


def tuning_restore(arr):
    _lib.check(_lib.load().m2h_tuning_restore(arr, _lib.TUNING_KNOBS), "m2h_tuning_restore")


class tuning_scope:
    """``with ops.tuning_scope(snapshot):`` -- the calling thread dispatches with the given knobs inside the block, with its own after."""

    def __init__(self, arr):
        self.arr = arr

    def __enter__(self):
        self.prev = tuning_snapshot()
        tuning_restore(self.arr)

    def __exit__(self, *exc):
        tuning_restore(self.prev)
        return False


MATH_FP32, MATH_BF16X3, MATH_BF16 = 0, 1, 2   # include/m2h.h M2H_MATH_*
_tls = threading.local()   # the calling thread's arithmetic (mirrors libm2h's thread-local m2h_set_math_mode)
FMT_SRC_SPLIT, FMT_W_SPLIT, FMT_DST_SPLIT, FMT_MATH_BF16X3, FMT_MATH_FP32 = 1, 2, 4, 8, 16   # include/m2h.h M2H_FMT_*


def pack_strip_conv1(w):
    """torch Conv2d weight [64][Ci >= 32][4][4] of the first encoder stage -> the strip kernel's register image (csrc/conv_strip.hip)."""
    _chk(w, "pack_strip_conv1")
    if w.dim() != 4 or w.shape[0] != 64 or w.shape[1] < 32 or tuple(w.shape[2:]) != (4, 4):
        raise RuntimeError("m2h.pack_strip_conv1: expected a [64, >=32, 4, 4] weight, got %s" % (tuple(w.shape),))
    lib = _lib.load()
    out = torch.empty(lib.m2h_strip_conv1_weight_bytes() // 4, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        _lib.check(lib.m2h_pack_strip_conv1(_ptr(w), int(w.shape[1]), _ptr(out), _stream(w)), "m2h_pack_strip_conv1")
    return out


def strip_conv1_fwd(mix, masks, wreg, scale, shift, cls_table=None, cls_val=None, slope=0.2):
    """Slice (+ bin2mono pre-op) + first encoder stage in one strip-walker launch: mix / masks [B,512,T,2] -> split32 NHWC
    [B,16,T/2,64] (bf16x3 arithmetic; include/m2h.h m2h_strip_conv1_fwd)."""
    _chk(mix, "strip_conv1_fwd")
    B, F, T, C = mix.shape
    if C != 2:
        raise RuntimeError("m2h.strip_conv1_fwd: expected [B,512,T,2]")
    out = torch.empty((B, 16, T // 2, 64), device=mix.device, dtype=torch.float32)
    with torch.cuda.device(mix.device):
        _lib.check(_lib.load().m2h_strip_conv1_fwd(_ptr(mix), _ptr(masks), _ptr(wreg), _ptr(scale), _ptr(shift), _ptr(cls_table), _ptr(cls_val),
                                                   _ptr(out), B, F, T, float(slope), _stream(mix)), "m2h_strip_conv1_fwd")
    return out


def strip_last_fwd(x, skip, wp_split32, scale, shift, head_w, head_b, Co):
    """Last decoder stage + head + de-slice in one strip-walker launch: x, skip split32 NHWC [B,H,W,64] -> BHWC
    [B,32H,2W,Co/16] (bf16x3 arithmetic; include/m2h.h m2h_strip_last_fwd)."""
    _chk(x, "strip_last_fwd")
    _chk(skip, "strip_last_fwd(skip)")
    B, H, W, C = x.shape
    if C != 64 or tuple(skip.shape) != tuple(x.shape):
        raise RuntimeError("m2h.strip_last_fwd: expected two [B,H,W,64] tensors")
    out = torch.empty((B, 32 * H, 2 * W, Co // 16), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().m2h_strip_last_fwd(_ptr(x), _ptr(skip), _ptr(wp_split32), _ptr(scale), _ptr(shift), _ptr(head_w), _ptr(head_b),
                                                  _ptr(out), B, H, W, int(Co), _stream(x)), "m2h_strip_last_fwd")
    return out


def split32(t):
    """fp32 tensor (innermost dimension a multiple of 32) -> same-shape tensor in the split32 layout of include/m2h.h."""
    _chk(t, "split32")
    if t.shape[-1] % 32 != 0:
        raise RuntimeError("m2h.split32: innermost dimension must be a multiple of 32")
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().m2h_split32(_ptr(t), _ptr(out), t.numel(), _stream(t)), "m2h_split32")
    return out


def set_math_mode(mode):
    """Arithmetic of the igemm forward engine for the CALLING THREAD (m2h_set_math_mode; no process-wide state: two threads may
    run different modes side by side, tests/test_gpu_unet.py).  Backward passes run on autograd's threads: the autograd Functions
    of m2h.functional carry the mode of their forward with them (functional.carries_math_mode).
    MATH_FP32 (default): fp32 matrix instructions, exact fp32 products.  MATH_BF16X3: fp32 operands split into bf16 hi + lo
    inside the kernel, products hi*hi + hi*lo + lo*hi on the bf16 matrix pipe with fp32 accumulation (~16 mantissa bits per
    product; tensors in HBM stay fp32).  Applies to shapes the scalar loader takes (channel counts multiples of 32)."""
    if mode not in (MATH_FP32, MATH_BF16X3, MATH_BF16):
        raise ValueError("math mode must be ops.MATH_FP32, ops.MATH_BF16X3 or ops.MATH_BF16")
    _lib.check(_lib.load().m2h_set_math_mode(int(mode)), "m2h_set_math_mode")
    _tls.math_mode = mode


def math_mode():
    return getattr(_tls, "math_mode", MATH_FP32)


class math_scope:
    """``with ops.math_scope(mode):`` -- the calling thread computes in `mode` inside the block and in its previous mode after."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = math_mode()
        if self.mode != self.prev:
            set_math_mode(self.mode)

    def __exit__(self, *exc):
        if self.mode != self.prev:
            set_math_mode(self.prev)
        return False


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("m2h.%s: tensor must live on the GPU (got %s); the m2h ops have no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("m2h.%s: expected %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("m2h.%s: tensor must be contiguous" % name)


def sep_slice_input(mix, masks=None):
    """K1/K2.  mix, masks: BHWC [B,F,T,C] -> NHWC [B,F/16,T,16*C]  (separator_cnn.py:73-90)."""
    _chk(mix, "sep_slice_input(mix)")
    _chk(masks, "sep_slice_input(masks)")
    B, F, T, C = mix.shape
    if masks is not None and masks.shape != mix.shape:
        raise RuntimeError("m2h.sep_slice_input: masks %s vs mix %s" % (tuple(masks.shape), tuple(mix.shape)))
    out = torch.empty((B, F // 16, T, 16 * C), device=mix.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(mix.device):
        _timed("sep_slice_input", {"bytes": (2 if masks is None else 3) * mix.numel() * 4}, mix.device,
               lambda: _lib.check(lib.m2h_sep_slice_input(_ptr(mix), _ptr(masks), _ptr(out), B, F, T, C, _stream(mix)),
                                  "m2h_sep_slice_input"))
    return out


def pack_conv_weight(w, ci_used=None):
    """[Co,Ci,KH,KW] -> [Co, KH*KW*ci_used] (tap-major, channel fastest)."""
    _chk(w, "pack_conv_weight")
    Co, Ci, KH, KW = w.shape
    cu = Ci if ci_used is None else ci_used
    wp = torch.empty((Co, KH * KW * cu), device=w.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(w.device):
        _lib.check(lib.m2h_pack_conv_weight(_ptr(w), _ptr(wp), Co, Ci, KH, KW, cu, _stream(w)), "m2h_pack_conv_weight")
    return wp


def pack_convT_weight(w):
    """ConvTranspose2d(4,2,1) weight [Ci,Co,4,4] -> [4 phases, Co, 4*Ci]."""
    _chk(w, "pack_convT_weight")
    Ci, Co, KH, KW = w.shape
    if (KH, KW) != (4, 4):
        raise RuntimeError("m2h.pack_convT_weight: only 4x4 kernels (got %dx%d)" % (KH, KW))
    wp = torch.empty((4, Co, 4 * Ci), device=w.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(w.device):
        _lib.check(lib.m2h_pack_convT_weight(_ptr(w), _ptr(wp), Ci, Co, _stream(w)), "m2h_pack_convT_weight")
    return wp


def unet_class_table(w, plane):
    _chk(w, "unet_class_table")
    Co, Ci, KH, KW = w.shape
    if (KH, KW) != (4, 4):
        raise RuntimeError("m2h.unet_class_table: only 4x4 kernels")
    table = torch.empty((9, Co), device=w.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(w.device):
        _lib.check(lib.m2h_unet_class_table(_ptr(w), _ptr(table), Co, Ci, plane, _stream(w)), "m2h_unet_class_table")
    return table


def fold_bn(gamma, beta, mean, var, eps):
    for t in (gamma, beta, mean, var):
        _chk(t, "fold_bn")
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=torch.float32)
    shift = torch.empty(C, device=gamma.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(gamma.device):
        _lib.check(lib.m2h_fold_bn(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps), _ptr(scale), _ptr(shift), C,
                                   _stream(gamma)), "m2h_fold_bn")
    return scale, shift


def unet_down_fwd(x, wp, scale, shift, Co, cls_table=None, cls_val=None):
    """K3.  x NHWC [B,H,W,Ci] -> NHWC [B,H/2,W/2,Co]."""
    _chk(x, "unet_down_fwd(x)")
    for t in (wp, scale, shift, cls_table, cls_val):
        _chk(t, "unet_down_fwd")
    B, H, W, Ci = x.shape
    if wp.numel() != Co * 16 * Ci:
        raise RuntimeError("m2h.unet_down_fwd: packed weight has %d elements, expected %d" % (wp.numel(), Co * 16 * Ci))
    if scale.numel() != Co or shift.numel() != Co:
        raise RuntimeError("m2h.unet_down_fwd: scale/shift size")
    if cls_table is not None and (cls_table.numel() != 9 * Co or cls_val.numel() != B):
        raise RuntimeError("m2h.unet_down_fwd: class table/val size")
    y = torch.empty((B, H // 2, W // 2, Co), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        M = B * (H // 2) * (W // 2)
        ws, wsb = _workspace(lib.m2h_unet_down_workspace_bytes(B, H, W, Ci, Co), x.device)
        meta = {"kernel": igemm_config(Co), "M": M, "N": Co, "K": 16 * Ci,
                "flops": 2.0 * M * Co * 16 * (Ci + (1 if cls_table is not None else 0)),
                "bytes": 4.0 * (x.numel() + y.numel() + wp.numel())}
        _timed("unet_down_fwd", meta, x.device,
               lambda: _lib.check(lib.m2h_unet_down_fwd(_ptr(x), _ptr(wp), _ptr(scale), _ptr(shift), _ptr(cls_table), _ptr(cls_val),
                                                        _ptr(y), B, H, W, Ci, Co, _ptr(ws), wsb, _stream(x)), "m2h_unet_down_fwd"))
    return y


def unet_up_fwd(x, skip, wp, scale, shift, Co):
    """K4.  x [B,H,W,C0] (+ skip [B,H,W,C1]) -> NHWC [B,2H,2W,Co]."""
    _chk(x, "unet_up_fwd(x)")
    _chk(skip, "unet_up_fwd(skip)")
    for t in (wp, scale, shift):
        _chk(t, "unet_up_fwd")
    B, H, W, C0 = x.shape
    C1 = 0
    if skip is not None:
        if skip.shape[:3] != x.shape[:3]:
            raise RuntimeError("m2h.unet_up_fwd: skip %s vs x %s" % (tuple(skip.shape), tuple(x.shape)))
        C1 = skip.shape[3]
    if wp.numel() != 4 * Co * 4 * (C0 + C1):
        raise RuntimeError("m2h.unet_up_fwd: packed weight has %d elements, expected %d" % (wp.numel(), 16 * Co * (C0 + C1)))
    if scale.numel() != Co or shift.numel() != Co:
        raise RuntimeError("m2h.unet_up_fwd: scale/shift size")
    y = torch.empty((B, 2 * H, 2 * W, Co), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        M = B * H * W
        ws, wsb = _workspace(lib.m2h_unet_up_workspace_bytes(B, H, W, C0, C1, Co), x.device)
        meta = {"kernel": igemm_config(Co), "M": 4 * M, "N": Co, "K": 4 * (C0 + C1), "flops": 2.0 * 4 * M * Co * 4 * (C0 + C1),
                "bytes": 4.0 * (x.numel() + (skip.numel() if skip is not None else 0) + y.numel() + wp.numel())}
        _timed("unet_up_fwd", meta, x.device,
               lambda: _lib.check(lib.m2h_unet_up_fwd(_ptr(x), _ptr(skip), _ptr(wp), _ptr(scale), _ptr(shift), _ptr(y), B, H, W,
                                                      C0, C1, Co, _ptr(ws), wsb, _stream(x)), "m2h_unet_up_fwd"))
    return y


def unet_head_fwd(x, wp, bias, Co):
    """K5.  x NHWC [B,H,W,Ci] -> BHWC [B,16*H,W,Co/16]."""
    _chk(x, "unet_head_fwd(x)")
    _chk(wp, "unet_head_fwd(wp)")
    _chk(bias, "unet_head_fwd(bias)")
    B, H, W, Ci = x.shape
    if wp.numel() != Co * Ci or bias.numel() != Co or Co % 16 != 0:
        raise RuntimeError("m2h.unet_head_fwd: bad weight/bias size")
    out = torch.empty((B, 16 * H, W, Co // 16), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        M = B * H * W
        meta = {"kernel": igemm_config(Co), "M": M, "N": Co, "K": Ci, "flops": 2.0 * M * Co * Ci,
                "bytes": 4.0 * (x.numel() + out.numel() + wp.numel())}
        _timed("unet_head_fwd", meta, x.device,
               lambda: _lib.check(lib.m2h_unet_head_fwd(_ptr(x), _ptr(wp), _ptr(bias), _ptr(out), B, H, W, Ci, Co, _stream(x)),
                                  "m2h_unet_head_fwd"))
    return out


def conv_igemm_f32(**kw):
    """Generic implicit-GEMM conv (m2h_conv_igemm_f32); keyword names = fields of m2h_conv_args, tensors
    for the pointer fields."""
    a = _lib.ConvArgs()
    dev_t = kw["src0"]
    for name, _ in _lib.ConvArgs._fields_:
        v = kw.get(name, None)
        if name in ("src0", "src1", "wp", "scale", "shift", "cls_table", "cls_val", "dst", "workspace"):
            _chk(v, "conv_igemm_f32(%s)" % name)
            setattr(a, name, v.data_ptr() if v is not None else None)
        elif v is not None:
            setattr(a, name, v)
    lib = _lib.load()
    with torch.cuda.device(dev_t.device):
        _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), _stream(dev_t)), "m2h_conv_igemm_f32")


# ----------------------------------------------------------------------------------------------------------------
# generic conv / linear on the igemm engine, and the RL-path ops
# ----------------------------------------------------------------------------------------------------------------
def conv2d_nhwc(x, wp, n_out, kh, kw, stride=1, pad=0, bias=None, scale=None, slope=1.0, x2=None, deslice=False, name="conv2d", out=None,
                operand_format=0):
    """Conv2d over NHWC activations through m2h_conv_igemm_f32.  x [B,H,W,C0] (+ x2 [B,H,W,C1] concatenated on channels),
    wp packed [n_out, kh*kw*(C0+C1)], bias -> epilogue shift, slope: 1 none / 0 ReLU / 0.2 LeakyReLU.
    Returns NHWC [B,Ho,Wo,n_out], or the de-sliced BHWC [B,16*Ho,Wo,n_out/16] when deslice."""
    _chk(x, name + "(x)")
    _chk(x2, name + "(x2)")
    for t in (wp, bias, scale):
        _chk(t, name)
    B, H, W, C0 = x.shape
    C1 = x2.shape[3] if x2 is not None else 0
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if wp.numel() != n_out * kh * kw * (C0 + C1):
        raise RuntimeError("m2h.%s: packed weight has %d elements, expected %d" % (name, wp.numel(), n_out * kh * kw * (C0 + C1)))
    ldc = n_out
    if out is not None and out.dim() == 2 and not out.is_contiguous():
        # a column block of a wider row-major matrix (the policy's concatenated features, rl/ppo/policy.py:103): rows ldc floats apart
        if not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (B, n_out) and out.stride(1) == 1 and Ho * Wo == 1
                and not deslice and out.stride(0) >= n_out and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0):
            raise RuntimeError("m2h.%s: a strided out must be a [B, n_out] column block of a row-major fp32 matrix" % name)
        ldc = out.stride(0)
    elif out is not None:
        _chk(out, name + "(out)")
        if out.numel() != B * Ho * Wo * n_out:
            raise RuntimeError("m2h.%s: out has %d elements, expected %d" % (name, out.numel(), B * Ho * Wo * n_out))
    elif deslice:
        out = torch.empty((B, 16 * Ho, Wo, n_out // 16), device=x.device, dtype=torch.float32)
    else:
        out = torch.empty((B, Ho, Wo, n_out), device=x.device, dtype=torch.float32)
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), C0, C1
    a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, Ho, Wo
    a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = stride, kh, kw, 1, -pad, 1, -pad
    a.conv_transpose, a.wp, a.N = 0, wp.data_ptr(), n_out
    a.scale = scale.data_ptr() if scale is not None else None
    a.shift = bias.data_ptr() if bias is not None else None
    a.slope = float(slope)
    a.cls_table, a.cls_val = None, None
    a.dst, a.Ho, a.Wo, a.os, a.ph, a.pw, a.ldc = out.data_ptr(), Ho, Wo, 1, 0, 0, ldc
    a.out_mode = OUT_DESLICE if deslice else OUT_NHWC
    a.operand_format = int(operand_format)   # FMT_* bits: split32 operands / output (bf16x3 math only)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        ws, wsb = _workspace(lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)), x.device)
        a.workspace, a.workspace_bytes = (ws.data_ptr() if ws is not None else None), wsb
        M = B * Ho * Wo
        meta = {"kernel": igemm_config(n_out), "M": M, "N": n_out, "K": kh * kw * (C0 + C1), "flops": 2.0 * M * n_out * kh * kw * (C0 + C1),
                "bytes": 4.0 * (x.numel() + (x2.numel() if x2 is not None else 0) + out.numel() + wp.numel())}
        _timed(name, meta, x.device, lambda: _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), _stream(x)), "m2h_conv_igemm_f32"))
    return out


def linear(x, w, bias=None, slope=1.0, name="linear", out=None):
    """y = act(x W^T + b): x [M,K], w [N,K] (torch Linear layout == packed [N][K]); out: optional [M,N] destination."""
    M, K = x.shape
    y = conv2d_nhwc(x.reshape(M, 1, 1, K), w, w.shape[0], 1, 1, bias=bias, slope=slope, name=name, out=out)
    return y.reshape(M, w.shape[0])


def pack_conv_weight_ex(w4d, ci_used, ci_out, out=None):
    """out: optional previously packed buffer of the same shape to refresh in place (its address stays valid for HIP graphs)."""
    _chk(w4d, "pack_conv_weight_ex")
    Co, Ci, KH, KW = w4d.shape
    if out is not None and (tuple(out.shape) != (Co, KH * KW * ci_out) or out.device != w4d.device):
        raise RuntimeError("pack_conv_weight_ex: out buffer does not match the packed shape")
    wp = out if out is not None else torch.empty((Co, KH * KW * ci_out), device=w4d.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(w4d.device):
        _lib.check(lib.m2h_pack_conv_weight_ex(_ptr(w4d), _ptr(wp), Co, Ci, KH, KW, ci_used, ci_out, _stream(w4d)), "m2h_pack_conv_weight_ex")
    return wp


def slice_concat_input(a, b=None, mul=None, bscale=None, op=0, out=None):
    """AcousticMem / AudioCNN input glue (m2h_slice_concat_input).  a [B,F,T,Ca], b [B,F,T,Cb] -> NHWC [B,F/16,T,16*(Ca+Cb)] (into `out` if given)."""
    for t in (a, b, mul, bscale):
        _chk(t, "slice_concat_input")
    B, F, T, Ca = a.shape
    Cb = b.shape[3] if b is not None else 0
    if b is not None and b.shape[:3] != a.shape[:3]:
        raise RuntimeError("m2h.slice_concat_input: a %s vs b %s" % (tuple(a.shape), tuple(b.shape)))
    if mul is not None and mul.shape != a.shape:
        raise RuntimeError("m2h.slice_concat_input: mul shape")
    if bscale is not None and bscale.numel() != B:
        raise RuntimeError("m2h.slice_concat_input: bscale size")
    shape = (B, F // 16, T, 16 * (Ca + Cb))
    if out is None:
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    elif tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32 or out.device != a.device:
        raise RuntimeError("m2h.slice_concat_input: out must be a contiguous fp32 tensor of shape %s on the inputs' device" % (shape,))
    lib = _lib.load()
    with torch.cuda.device(a.device):
        _timed("slice_concat_input", {"bytes": 4.0 * (2 * out.numel())}, a.device,
               lambda: _lib.check(lib.m2h_slice_concat_input(_ptr(a), Ca, _ptr(b), Cb, _ptr(mul), _ptr(bscale), op, _ptr(out), B, F, T,
                                                             _stream(a)), "m2h_slice_concat_input"))
    return out


def visual_input(rgb, depth=None, out=None):
    _chk(rgb, "visual_input(rgb)")
    _chk(depth, "visual_input(depth)")
    B, H, W, C = rgb.shape
    if C != 3:
        raise RuntimeError("m2h.visual_input: rgb must have 3 channels")
    if out is None:
        out = torch.empty((B, H, W, 4), device=rgb.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, H, W, 4) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != rgb.device:
        raise RuntimeError("m2h.visual_input: out must be a contiguous fp32 [B, H, W, 4] tensor on the inputs' device")
    lib = _lib.load()
    with torch.cuda.device(rgb.device):
        _lib.check(lib.m2h_visual_input(_ptr(rgb), _ptr(depth), _ptr(out), B, H, W, _stream(rgb)), "m2h_visual_input")
    return out


def gru_gates(gi, gh_raw, bhh, hprev, mask=None, out=None):
    for t in (gi, gh_raw, bhh, hprev, mask, out):
        _chk(t, "gru_gates")
    M, H = hprev.shape
    if gi.shape != (M, 3 * H) or gh_raw.shape != (M, 3 * H) or bhh.numel() != 3 * H or (mask is not None and mask.numel() != M):
        raise RuntimeError("m2h.gru_gates: shape mismatch")
    hout = out if out is not None else torch.empty_like(hprev)
    lib = _lib.load()
    with torch.cuda.device(gi.device):
        _lib.check(lib.m2h_gru_gates(_ptr(gi), _ptr(gh_raw), _ptr(bhh), _ptr(hprev), _ptr(mask), _ptr(hout), M, H, _stream(gi)), "m2h_gru_gates")
    return hout


GRU_STEP_MAX_ROWS = 16


def gru_step(gi, whh, bhh, hprev, mask=None, gh_out=None, out=None):
    """One GRU time step for M <= 16 rows in one launch (m2h_gru_step): recurrent product + gates.  -> (hout, gh_raw)."""
    for t in (gi, whh, bhh, hprev, mask, gh_out, out):
        _chk(t, "gru_step")
    M, H = hprev.shape
    if gi.shape != (M, 3 * H) or whh.shape != (3 * H, H) or bhh.numel() != 3 * H or (mask is not None and mask.numel() != M):
        raise RuntimeError("m2h.gru_step: shape mismatch")
    hout = out if out is not None else torch.empty_like(hprev)
    gh = gh_out if gh_out is not None else torch.empty_like(gi)
    if hout.shape != (M, H) or gh.shape != (M, 3 * H):
        raise RuntimeError("m2h.gru_step: output shape mismatch")
    with torch.cuda.device(gi.device):
        _timed("gru.step", {"M": M, "N": 3 * H, "K": H}, gi.device,
               lambda: _lib.check(_lib.load().m2h_gru_step(_ptr(gi), _ptr(whh), _ptr(bhh), _ptr(hprev), _ptr(mask), _ptr(gh), _ptr(hout), M, H,
                                                           _stream(gi)), "m2h_gru_step"))
    return hout, gh


def gru_cell(x, wih, bih, whh, bhh, hprev, mask=None):
    """The whole GRU cell of a no-grad single step for M <= 16 rows in one launch (m2h_gru_cell) -> hout [M,H]."""
    for t in (x, wih, bih, whh, bhh, hprev, mask):
        _chk(t, "gru_cell")
    M, H = hprev.shape
    I = x.shape[1]
    if x.shape[0] != M or wih.shape != (3 * H, I) or whh.shape != (3 * H, H) or bih.numel() != 3 * H or bhh.numel() != 3 * H or \
            (mask is not None and mask.numel() != M):
        raise RuntimeError("m2h.gru_cell: shape mismatch")
    hout = torch.empty_like(hprev)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().m2h_gru_cell(_ptr(x), _ptr(wih), _ptr(bih), _ptr(whh), _ptr(bhh), _ptr(hprev), _ptr(mask), _ptr(hout), M, I, H,
                                            _stream(x)), "m2h_gru_cell")
    return hout


def policy_heads(feats, Wa, ba, Wc, bc, actions=None):
    """-> value [M,1], logp_all [M,A], probs [M,A], entropy [M], logp_act [M,1] or None."""
    for t in (feats, Wa, ba, Wc, bc):
        _chk(t, "policy_heads")
    _chk(actions, "policy_heads(actions)", torch.int64)
    M, H = feats.shape
    A = Wa.shape[0]
    dev = feats.device
    value = torch.empty((M, 1), device=dev)
    logp_all = torch.empty((M, A), device=dev)
    probs = torch.empty((M, A), device=dev)
    ent = torch.empty((M,), device=dev)
    logp_act = torch.empty((M, 1), device=dev) if actions is not None else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.m2h_policy_heads(_ptr(feats), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), _ptr(actions), _ptr(value), _ptr(logp_all),
                                        _ptr(probs), _ptr(ent), _ptr(logp_act), M, H, A, _stream(feats)), "m2h_policy_heads")
    return value, logp_all, probs, ent, logp_act


def policy_heads_act(feats, Wa, ba, Wc, bc, noise=None, rng=None, noise_out=None):
    """Heads + action + its log-probability in one launch (m2h_policy_heads_act): noise [M,A] Exp(1) -> the multinomial draw
    argmax(probs / noise); None -> the mode; rng = int64 device tensor [seed, counter] -> the noise is drawn inside the kernel
    (m2h_policy_heads_act_rng; the caller advances the counter; noise_out: optional [M,A] float tensor that receives the noise drawn).
    -> value [M,1], logp_all [M,A], probs [M,A], entropy [M], action [M,1] int64, logp_act [M,1]."""
    for t in (feats, Wa, ba, Wc, bc, noise):
        _chk(t, "policy_heads_act")
    M, H = feats.shape
    A = Wa.shape[0]
    dev = feats.device
    if noise is not None and tuple(noise.shape) != (M, A):
        raise RuntimeError("m2h.policy_heads_act: noise must be [%d, %d]" % (M, A))
    value = torch.empty((M, 1), device=dev)
    logp_all = torch.empty((M, A), device=dev)
    probs = torch.empty((M, A), device=dev)
    ent = torch.empty((M,), device=dev)
    action = torch.empty((M, 1), device=dev, dtype=torch.int64)
    logp_act = torch.empty((M, 1), device=dev)
    if rng is not None:
        if noise is not None or rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda or not rng.is_contiguous():
            raise RuntimeError("m2h.policy_heads_act: rng must be a contiguous int64 device tensor [seed, counter] (and noise None)")
        if noise_out is not None:
            _chk(noise_out, "policy_heads_act")
            if tuple(noise_out.shape) != (M, A):
                raise RuntimeError("m2h.policy_heads_act: noise_out must be [%d, %d]" % (M, A))
        with torch.cuda.device(dev):
            _lib.check(_lib.load().m2h_policy_heads_act_rng(_ptr(feats), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), _ptr(rng), _ptr(value),
                                                            _ptr(logp_all), _ptr(probs), _ptr(ent), _ptr(action), _ptr(logp_act), _ptr(noise_out),
                                                            M, H, A, _stream(feats)), "m2h_policy_heads_act_rng")
        return value, logp_all, probs, ent, action, logp_act
    if noise_out is not None:
        raise RuntimeError("m2h.policy_heads_act: noise_out records the fused draw's noise (rng); with caller-supplied noise there is nothing to record")
    with torch.cuda.device(dev):
        _lib.check(_lib.load().m2h_policy_heads_act(_ptr(feats), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), _ptr(noise), _ptr(value),
                                                    _ptr(logp_all), _ptr(probs), _ptr(ent), _ptr(action), _ptr(logp_act), M, H, A,
                                                    _stream(feats)), "m2h_policy_heads_act")
    return value, logp_all, probs, ent, action, logp_act


def gather_logp(logp_all, actions):
    _chk(logp_all, "gather_logp")
    _chk(actions, "gather_logp(actions)", torch.int64)
    M, A = logp_all.shape
    out = torch.empty((M, 1), device=logp_all.device)
    lib = _lib.load()
    with torch.cuda.device(logp_all.device):
        _lib.check(lib.m2h_gather_logp(_ptr(logp_all), _ptr(actions), _ptr(out), M, A, _stream(logp_all)), "m2h_gather_logp")
    return out


def sample_actions(probs, noise):
    """argmax(probs / noise) per row -> [M,1] int64: torch.multinomial(probs, 1, True) given its Exp(1) noise."""
    _chk(probs, "sample_actions")
    _chk(noise, "sample_actions(noise)")
    if noise.shape != probs.shape:
        raise RuntimeError("m2h sample_actions: noise %s does not match probs %s" % (tuple(noise.shape), tuple(probs.shape)))
    M, A = probs.shape
    out = torch.empty((M, 1), dtype=torch.int64, device=probs.device)
    lib = _lib.load()
    with torch.cuda.device(probs.device):
        _lib.check(lib.m2h_sample_actions(_ptr(probs), _ptr(noise), _ptr(out), M, A, _stream(probs)), "m2h_sample_actions")
    return out


def gae_returns(rewards, value_preds, masks, next_value, returns, use_gae, gamma, tau):
    """In place on value_preds[-1] and returns (RolloutStoragePol.compute_returns).  Shapes [T,N,1]/[T+1,N,1]/[N,1]."""
    for t in (rewards, value_preds, masks, next_value, returns):
        _chk(t, "gae_returns")
    T, N = rewards.shape[0], rewards.shape[1]
    lib = _lib.load()
    with torch.cuda.device(rewards.device):
        _lib.check(lib.m2h_gae_returns(_ptr(rewards), _ptr(value_preds), _ptr(masks), _ptr(next_value), _ptr(returns), T, N,
                                       1 if use_gae else 0, float(gamma), float(tau), _stream(rewards)), "m2h_gae_returns")


def advantages(returns, value_preds, mode, eps=1e-5):
    """returns/value_preds [T+1,N,1] -> (adv [T,N,1], stats [2]); mode 0 raw, 1 local-normalised, 2 raw + local mean."""
    _chk(returns, "advantages")
    _chk(value_preds, "advantages")
    T = returns.shape[0] - 1
    n = T * returns.shape[1]
    adv = torch.empty((T,) + tuple(returns.shape[1:]), device=returns.device)
    stats = torch.empty(2, device=returns.device)
    lib = _lib.load()
    with torch.cuda.device(returns.device):
        _lib.check(lib.m2h_advantages(_ptr(returns), _ptr(value_preds), _ptr(adv), _ptr(stats), n, mode, float(eps), _stream(returns)),
                   "m2h_advantages")
    return adv, stats


def adv_sqdiff(adv, gmean):
    out = torch.empty(1, device=adv.device)
    lib = _lib.load()
    with torch.cuda.device(adv.device):
        _lib.check(lib.m2h_adv_sqdiff(_ptr(adv), _ptr(gmean), _ptr(out), adv.numel(), _stream(adv)), "m2h_adv_sqdiff")
    return out


def adv_apply(adv, gmean, gvar, eps=1e-5):
    lib = _lib.load()
    with torch.cuda.device(adv.device):
        _lib.check(lib.m2h_adv_apply(_ptr(adv), _ptr(gmean), _ptr(gvar), adv.numel(), float(eps), _stream(adv)), "m2h_adv_apply")
    return adv


def ppo_loss(values, logp, old_values, returns, adv, old_logp, clip, value_loss_coef=1.0, use_clipped_value_loss=True, want_grads=False,
             entropy=None, entropy_coef=0.0):
    """-> (out[4] = (value_loss, action_loss, mean entropy, total_loss), g_values, g_logp).
    clip: python float, or a 1-element device tensor read when the kernel runs (HIP-graph replay of the update)."""
    clip_dev = clip if torch.is_tensor(clip) else None
    for t in (values, logp, old_values, returns, adv, old_logp, entropy, clip_dev):
        _chk(t, "ppo_loss")
    n = values.numel()
    out = torch.empty(4, device=values.device)
    gv = torch.empty_like(values) if want_grads else None
    gl = torch.empty_like(logp) if want_grads else None
    lib = _lib.load()
    with torch.cuda.device(values.device):
        _lib.check(lib.m2h_ppo_loss(_ptr(values), _ptr(logp), _ptr(old_values), _ptr(returns), _ptr(adv), _ptr(old_logp), _ptr(entropy),
                                    0.0 if clip_dev is not None else float(clip), _ptr(clip_dev), 1 if use_clipped_value_loss else 0, float(value_loss_coef), float(entropy_coef), _ptr(out),
                                    _ptr(gv), _ptr(gl), n, _stream(values)), "m2h_ppo_loss")
    return out, gv, gl


def sq_stats(pred, gt_comps, gt_off=0):
    """Per-env (sum (pred-gt)^2, sum gt^2); pred [N,F,T,1], gt_comps [N,F,T,Cg] read at channel gt_off."""
    _chk(pred, "sq_stats")
    _chk(gt_comps, "sq_stats")
    N = pred.shape[0]
    L = pred.numel() // N
    stats = torch.empty((N, 2), device=pred.device)
    lib = _lib.load()
    with torch.cuda.device(pred.device):
        _lib.check(lib.m2h_sq_stats(_ptr(pred), _ptr(gt_comps), gt_comps.shape[-1], gt_off, _ptr(stats), N, L, _stream(pred)), "m2h_sq_stats")
    return stats


def rewards_from_stats(next_stats, cur_stats, not_done, L, quality_improvement, mult=10.0):
    N = next_stats.shape[0]
    rewards = torch.empty((N, 1), device=next_stats.device)
    lib = _lib.load()
    with torch.cuda.device(next_stats.device):
        _lib.check(lib.m2h_rewards_from_stats(_ptr(next_stats), _ptr(cur_stats), _ptr(not_done), _ptr(rewards), N, L,
                                              1 if quality_improvement else 0, float(mult), _stream(next_stats)), "m2h_rewards_from_stats")
    return rewards


def stft_l2(pred, gt_comps, nch, mix=None):
    """STFT-L2 per env -> [N,1].  pred [N,F,T,Cp], gt_comps [N,F,T,Cg]; mix given => pred is a mask on exp(mix)-1."""
    _chk(pred, "stft_l2")
    _chk(gt_comps, "stft_l2")
    _chk(mix, "stft_l2")
    N = pred.shape[0]
    L = pred.shape[1] * pred.shape[2]
    out = torch.empty((N, 1), device=pred.device)
    lib = _lib.load()
    with torch.cuda.device(pred.device):
        _lib.check(lib.m2h_stft_l2(_ptr(mix), _ptr(pred), pred.shape[3], _ptr(gt_comps), gt_comps.shape[3], nch, 1 if mix is not None else 0,
                                   _ptr(out), N, L, _stream(pred)), "m2h_stft_l2")
    return out


def bss_metrics(ref, est, mix_l, mix_r=None):
    """Waveform metrics of eval_metrics.scale_bss_eval per clip: ref/est/mix [S, L] -> [S, 11] (order: eval_metrics.METRIC_ORDER)."""
    for t in (ref, est, mix_l, mix_r):
        _chk(t, "bss_metrics")
    S, L = ref.shape
    if est.shape != ref.shape or mix_l.shape != ref.shape or (mix_r is not None and mix_r.shape != ref.shape):
        raise RuntimeError("m2h.bss_metrics: all waveforms must be [S, L]")
    out = torch.empty((S, 11), device=ref.device)
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        _lib.check(lib.m2h_bss_metrics(_ptr(ref), _ptr(est), _ptr(mix_l), _ptr(mix_r), _ptr(out), S, L, _stream(ref)), "m2h_bss_metrics")
    return out


def gather_envs(src, perm):
    """src [T,N,...] , perm [Nsel] int64 (device) -> [T*Nsel, ...]  (recurrent_generator stack + flatten)."""
    if not src.is_cuda or not src.is_contiguous():
        raise RuntimeError("m2h.gather_envs: src must be a contiguous GPU tensor")
    _chk(perm, "gather_envs(perm)", torch.int64)
    T, N = src.shape[0], src.shape[1]
    nsel = perm.numel()
    row = src[0, 0].numel() * src.element_size()
    dst = torch.empty((T * nsel,) + tuple(src.shape[2:]), device=src.device, dtype=src.dtype)
    lib = _lib.load()
    with torch.cuda.device(src.device):
        _timed("gather_envs", {"bytes": 2.0 * dst.numel() * dst.element_size()}, src.device,
               lambda: _lib.check(lib.m2h_gather_envs(_ptr(src), _ptr(perm), _ptr(dst), T, N, nsel, row, _stream(src)), "m2h_gather_envs"))
    return dst


def episode_stats_update(stats, rewards, dist_probs, bin_losses, mono_losses, monoFromMem_losses, not_done, ndgs=None, dgs=None):
    """The per-episode bookkeeping of one rollout step (ppo_trainer.py:407-478) in one launch.  stats: object with the seventeen
    [N,1] / [N,A] float tensors named in _lib.EPISODE_STATS_FIELDS, updated in place.  ndgs / dgs: the env's distance infos of
    this step ([N,1]; None = zeros)."""
    st = _lib.EpisodeStats()
    for name in _lib.EPISODE_STATS_FIELDS:
        t = getattr(stats, name)
        _chk(t, "episode_stats_update")
        setattr(st, name, t.data_ptr())
    ins = [t.contiguous() for t in (rewards, dist_probs, bin_losses, mono_losses, monoFromMem_losses, not_done)]
    for t in ins:
        _chk(t, "episode_stats_update")
    N, A = ins[1].shape
    if any(t.numel() != N for t in ins[:1] + ins[2:]) or stats.episode_dist_probs.numel() != N * A or stats.episode_rewards.numel() != N:
        raise RuntimeError("m2h.episode_stats_update: per-env tensors must hold one value per env (dist_probs: [N, A])")
    extra = []
    for t in (ndgs, dgs):
        if t is not None:
            t = t.contiguous()
            _chk(t, "episode_stats_update")
            if t.numel() != N:
                raise RuntimeError("m2h.episode_stats_update: ndgs / dgs must hold one value per env")
        extra.append(t)
    with torch.cuda.device(ins[0].device):
        _lib.check(_lib.load().m2h_episode_stats_update(ctypes.byref(st), *[_ptr(t) for t in ins], *[_ptr(t) if t is not None else None for t in extra],
                                                        N, A, _stream(ins[0])), "m2h_episode_stats_update")


def pack_batch(items):
    """Batched weight packing (m2h_pack_batch): items = (kind, src weight tensor, dst packed tensor, p[6]); one launch per 48."""
    if not items:
        return
    lib = _lib.load()
    dev = items[0][1].device
    for i0 in range(0, len(items), _lib.PACK_BATCH_MAX):
        chunk = items[i0:i0 + _lib.PACK_BATCH_MAX]
        arr = (_lib.PackItem * len(chunk))()
        for j, (kind, src, dst, prm) in enumerate(chunk):
            _chk(src, "pack_batch(src)")
            _chk(dst, "pack_batch(dst)")
            arr[j].src, arr[j].dst, arr[j].kind = src.data_ptr(), dst.data_ptr(), int(kind)
            for k in range(6):
                arr[j].p[k] = int(prm[k])
        with torch.cuda.device(dev):
            _lib.check(lib.m2h_pack_batch(arr, len(chunk), _stream(items[0][1])), "m2h_pack_batch")


_step_stats_scratch = {}


def step_stats_scratch(N, dev):
    """Scratch of m2h_rollout_step_stats for N envs: (partial sums, zeroed tickets).  The kernel leaves the tickets at zero, so one
    allocation serves every later launch of its owner; a trainer makes its own at setup (outside any HIP-graph capture: the zeroing
    is then not a node of the captured step) and passes it in."""
    return (torch.empty(_lib.load().m2h_step_stats_workspace_bytes(N) // 4, device=dev), torch.zeros(N, dtype=torch.int32, device=dev))


def rollout_step_stats(stats, next_mem, next_gt_mono_comps, mem, gt_mono_comps, masks, mix, gt_bin_comps, mono, not_done, probs,
                       env_rewards=None, ndgs=None, dgs=None, override=True, extra=False, extra_mult=10.0, scratch=None):
    """The per-env bookkeeping of one rollout step in ONE launch (m2h_rollout_step_stats): reward, the three STFT-L2 distances and
    the per-episode statistics update (ppo_trainer.py:375-455).  Returns (rewards [N,1], losses [3,N]: bin / mono / mono-from-memory);
    `stats` (the object of episode_stats_update) is updated in place.  Scratch (partial sums, tickets) is cached per (device, stream, N)."""
    a = _lib.StepStatsArgs()
    N, A = probs.shape
    L = mem.shape[1] * mem.shape[2]
    tens = dict(next_mem=next_mem if override else None, next_gt_mono_comps=next_gt_mono_comps if override else None, mem=mem,
                gt_mono_comps=gt_mono_comps, masks=masks, mix=mix, gt_bin_comps=gt_bin_comps, mono=mono, not_done=not_done,
                env_rewards=None if override else env_rewards, probs=probs, ndgs=ndgs, dgs=dgs)
    keep = []
    for name, t in tens.items():
        if t is not None:
            t = t.contiguous()
            _chk(t, "rollout_step_stats")
            keep.append(t)
            setattr(a, name, t.data_ptr())
    if mem.numel() != N * L or masks.numel() != 2 * N * L or mix.numel() != 2 * N * L or gt_mono_comps.numel() != 4 * N * L or \
            gt_bin_comps.numel() != 8 * N * L or mono.numel() != N * L or not_done.numel() != N:
        raise RuntimeError("m2h.rollout_step_stats: tensor sizes do not match N = %d envs x L = %d bins" % (N, L))
    for name in _lib.EPISODE_STATS_FIELDS:
        t = getattr(stats, name)
        _chk(t, "rollout_step_stats")
        setattr(a.stats, name, t.data_ptr())
    dev = mem.device
    # the last-arriver protocol of the kernel assumes exclusive use of its tickets and partial slabs while a launch is in flight:
    # launches on ONE stream are ordered, so the scratch is per (device, stream, N) -- two trainers, or a graph replay beside an
    # eager step on another stream, never share it (a captured graph holds the addresses of the scratch of its capture stream)
    if scratch is None:     # callers without a scratch of their own
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, N)
        scratch = _step_stats_scratch.get(key)
        if scratch is None:
            scratch = _step_stats_scratch[key] = step_stats_scratch(N, dev)
    rewards = torch.empty((N, 1), device=dev)
    losses = torch.empty((3, N), device=dev)
    a.rewards, a.losses, a.partial, a.tickets = rewards.data_ptr(), losses.data_ptr(), scratch[0].data_ptr(), scratch[1].data_ptr()
    a.N, a.L, a.A, a.override_rewards, a.extra_reward, a.extra_mult = N, L, A, int(bool(override)), int(bool(extra)), float(extra_mult)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().m2h_rollout_step_stats(ctypes.byref(a), _stream(mem)), "m2h_rollout_step_stats")
    return rewards, losses


def rows_copy(items, idx):
    """Batched row copies with device-resident row indices (m2h_rows_copy).  items: (src, dst, src_slot, dst_slot) with
    tensors; a slot >= 0 selects row idx[slot] along dim 0 of that tensor (the other side is then one whole row), a negative
    slot means the tensor itself is the row.  idx: int64 device tensor."""
    lib = _lib.load()
    if idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous():
        raise RuntimeError("m2h.rows_copy: idx must be a contiguous int64 device tensor")
    arr = []
    for src, dst, ss, ds in items:
        for t in (src, dst):
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("m2h.rows_copy: tensors must be contiguous device tensors")
        if src.dtype != dst.dtype:
            raise RuntimeError("m2h.rows_copy: dtype mismatch %s -> %s" % (src.dtype, dst.dtype))
        srow = src[0] if ss >= 0 else src
        drow = dst[0] if ds >= 0 else dst
        if srow.numel() != drow.numel():
            raise RuntimeError("m2h.rows_copy: row sizes differ (%s vs %s)" % (tuple(srow.shape), tuple(drow.shape)))
        if max(ss, ds) >= idx.numel():
            raise RuntimeError("m2h.rows_copy: slot out of range")
        arr.append(_lib.RowCopy(src.data_ptr(), dst.data_ptr(), srow.numel() * srow.element_size(), ss, ds))
    with torch.cuda.device(idx.device):
        for i in range(0, len(arr), _lib.ROWS_COPY_MAX):
            chunk = arr[i:i + _lib.ROWS_COPY_MAX]
            _lib.check(lib.m2h_rows_copy((_lib.RowCopy * len(chunk))(*chunk), len(chunk), _ptr(idx), _stream(idx)), "m2h_rows_copy")


def step_index_advance(idx, t_pol, t_sep, rng=None, rng_inc=0):
    """(pol_step, pol_step + 1, sep_step + 1) -> the next step's, on the device (m2h_step_index_advance); rng: the fused sampler's
    [seed, counter] state, whose counter moves on by rng_inc in the same launch."""
    if idx.dtype != torch.int64 or not idx.is_cuda or idx.numel() != 3 or not idx.is_contiguous():
        raise RuntimeError("m2h.step_index_advance: idx must be a contiguous int64 device tensor of 3 elements")
    with torch.cuda.device(idx.device):
        if rng is not None:
            _lib.check(_lib.load().m2h_step_index_advance_rng(_ptr(idx), int(t_pol), int(t_sep), _ptr(rng), int(rng_inc), _stream(idx)), "m2h_step_index_advance_rng")
            return
        _lib.check(_lib.load().m2h_step_index_advance(_ptr(idx), int(t_pol), int(t_sep), _stream(idx)), "m2h_step_index_advance")


def synth_env_step(actions, node, angle, n_nodes):
    """In-place pose update of the synthetic env (m2h_synth_env_step).  actions / node / angle: int64 [N] device tensors."""
    for t in (actions, node, angle):
        _chk(t, "synth_env_step", torch.int64)
    N = node.numel()
    if actions.numel() != N or angle.numel() != N:
        raise RuntimeError("m2h.synth_env_step: one action, node and angle per env")
    with torch.cuda.device(node.device):
        _lib.check(_lib.load().m2h_synth_env_step(_ptr(actions), _ptr(node), _ptr(angle), int(n_nodes), N, _stream(node)), "m2h_synth_env_step")


def synth_env_observe(pools, node, angle, audio_idx):
    """The synthetic env's observation lookup in one launch (m2h_synth_env_observe).  pools: list of (tensor [P, ...], kind) with
    kind 0 = frames indexed by node * 4 + angle, 1 = audio pool indexed by audio_idx.  -> list of gathered [N, ...] tensors."""
    for t in (node, angle, audio_idx):
        _chk(t, "synth_env_observe", torch.int64)
    N = node.numel()
    outs, arr = [], []
    for src, kind in pools:
        if not src.is_cuda or not src.is_contiguous():
            raise RuntimeError("m2h.synth_env_observe: pools must be contiguous device tensors")
        dst = torch.empty((N,) + tuple(src.shape[1:]), device=src.device, dtype=src.dtype)
        outs.append(dst)
        arr.append(_lib.RowCopy(src.data_ptr(), dst.data_ptr(), src[0].numel() * src.element_size(), int(kind), -1))
    with torch.cuda.device(node.device):
        _lib.check(_lib.load().m2h_synth_env_observe((_lib.RowCopy * len(arr))(*arr), len(arr), _ptr(node), _ptr(angle), _ptr(audio_idx), N,
                                                     _stream(node)), "m2h_synth_env_observe")
    return outs


def take_envs(src, perm, identity=False):
    """gather_envs, or -- when the caller knows the selection is ALL environments in storage order -- the same rows as a
    zero-copy view [T*N, ...] of the storage."""
    if identity:
        return src.reshape((src.shape[0] * src.shape[1],) + tuple(src.shape[2:]))
    return gather_envs(src, perm)


def bin_l1_loss(mix, masks, gt, cstep=2, want_grad=False):
    """mean |(exp(mix)-1)*masks - gt[..., cstep*c]| (ppo.py:219-221 with cstep 2 on gt_bin_comps; passive_trainer.py:270-272 with
    cstep 1 on gt_bin_mag) -> 0-dim device tensor (and d loss / d masks when want_grad)."""
    for t in (mix, masks, gt):
        _chk(t, "bin_l1_loss")
    npix = mix.numel() // 2
    loss = torch.empty(1, device=mix.device)
    grad = torch.empty_like(masks) if want_grad else None
    scratch = torch.empty(1024, device=mix.device)
    lib = _lib.load()
    with torch.cuda.device(mix.device):
        _lib.check(lib.m2h_bin_l1_loss(_ptr(mix), _ptr(masks), _ptr(gt), gt.shape[-1], cstep, _ptr(loss), _ptr(grad), _ptr(scratch), npix,
                                       _stream(mix)), "m2h_bin_l1_loss")
    return (loss[0], grad) if want_grad else loss[0]


def l1_loss_nhwc16(y, gt_comps, off=0, want_grad=True):
    """F.l1_loss(deslice(y), gt_comps[..., off:off+1]) for y still in its conv's NHWC layout [B, 32, T, 16] (m2h_l1_loss_nhwc16) ->
    (0-dim loss, d loss / d y in the same layout or None)."""
    _chk(y, "l1_loss_nhwc16")
    _chk(gt_comps, "l1_loss_nhwc16")
    B, H, T, C = y.shape
    if H != 32 or C != 16 or tuple(gt_comps.shape[:3]) != (B, 512, T):
        raise RuntimeError("m2h.l1_loss_nhwc16: y %s must be [B, 32, T, 16] and gt_comps %s [B, 512, T, *]" % (tuple(y.shape), tuple(gt_comps.shape)))
    loss = torch.empty(1, device=y.device)
    dy = torch.empty_like(y) if want_grad else None
    scratch = torch.empty(32 * B, device=y.device)
    lib = _lib.load()
    with torch.cuda.device(y.device):
        _lib.check(lib.m2h_l1_loss_nhwc16(_ptr(y), _ptr(gt_comps), gt_comps.shape[-1], int(off), _ptr(loss), _ptr(dy), _ptr(scratch), B, T, _stream(y)),
                   "m2h_l1_loss_nhwc16")
    return loss[0], dy


def conv3x3_l1_supported(h):
    """True when m2h_conv3x3_l1_nhwc16 takes h [B, H, T, C] (the image-row kernels' shapes: 32 x 32 x 32, B >= 64)."""
    B, H, T, C = h.shape
    return bool(_lib.load().m2h_conv3x3_l1_nhwc16_supported(B, H, T, C))


def conv3x3_l1_nhwc16(h, wp, gt_plane):
    """F.l1_loss(deslice(conv3x3(h, w)), gt) in one launch (m2h_conv3x3_l1_nhwc16): h NHWC [B, 32, 32, 32], wp packed [16, 288], gt_plane
    [B, 512, 32, 1] contiguous -> (0-dim loss, d loss / d y in the conv's NHWC layout [B, 32, 32, 16]).  The conv's output is never stored."""
    for t in (h, wp, gt_plane):
        _chk(t, "conv3x3_l1_nhwc16")
    B, H, T, C = h.shape
    if tuple(gt_plane.shape) not in ((B, 16 * H, T, 1), (B, 16 * H, T)) or wp.numel() != 16 * 9 * C:
        raise RuntimeError("m2h.conv3x3_l1_nhwc16: gt_plane %s must be [B, 16 H, T, 1] and wp [16, 9 C]" % (tuple(gt_plane.shape),))
    dy = torch.empty((B, H, T, 16), device=h.device)
    loss = torch.empty(1, device=h.device)
    partials = torch.empty(1024, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(_lib.load().m2h_conv3x3_l1_nhwc16(_ptr(h), _ptr(wp), _ptr(gt_plane), _ptr(dy), _ptr(loss), _ptr(partials), B, H, T, C, _stream(h)),
                   "m2h_conv3x3_l1_nhwc16")
    return loss[0], dy


def sep_slice_input_plane(mix, cls_val, ldo=36):
    """binSep stage-0 training input: slice + (target_class+1) plane as channel 32, zero padded to ldo channels."""
    _chk(mix, "sep_slice_input_plane")
    _chk(cls_val, "sep_slice_input_plane")
    B, F, T, C = mix.shape
    out = torch.empty((B, F // 16, T, ldo), device=mix.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(mix.device):
        _lib.check(lib.m2h_sep_slice_input_plane(_ptr(mix), _ptr(cls_val), _ptr(out), B, F, T, C, ldo, _stream(mix)), "m2h_sep_slice_input_plane")
    return out


def unet_up_fwd_raw(x, skip, wp, Co):
    """Transposed conv 4x4/s2/p1 of cat(x, skip) WITHOUT BatchNorm/activation (training path: batch statistics come next)."""
    _chk(x, "unet_up_fwd_raw(x)")
    _chk(skip, "unet_up_fwd_raw(skip)")
    _chk(wp, "unet_up_fwd_raw(wp)")
    B, H, W, C0 = x.shape
    C1 = skip.shape[3] if skip is not None else 0
    if wp.numel() != 16 * Co * (C0 + C1):
        raise RuntimeError("m2h.unet_up_fwd_raw: packed weight size")
    y = torch.empty((B, 2 * H, 2 * W, Co), device=x.device, dtype=torch.float32)
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), (skip.data_ptr() if skip is not None else None), C0, C1
    a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, H, W
    a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = 1, 2, 2, 0, 0, 0, 0
    a.conv_transpose, a.wp, a.N = 1, wp.data_ptr(), Co
    a.scale, a.shift, a.slope, a.cls_table, a.cls_val = None, None, 1.0, None, None
    a.dst, a.Ho, a.Wo, a.os, a.ph, a.pw, a.ldc, a.out_mode = y.data_ptr(), 2 * H, 2 * W, 2, 0, 0, Co, OUT_NHWC
    lib = _lib.load()
    with torch.cuda.device(x.device):
        ws, wsb = _workspace(lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)), x.device)
        a.workspace, a.workspace_bytes = (ws.data_ptr() if ws is not None else None), wsb
        M = B * H * W
        meta = {"kernel": igemm_config(Co), "M": 4 * M, "N": Co, "K": 4 * (C0 + C1), "flops": 2.0 * 4 * M * Co * 4 * (C0 + C1)}
        _timed("unet_up_fwd_raw", meta, x.device, lambda: _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), _stream(x)), "m2h_conv_igemm_f32"))
    return y


def unet_up_head_fwd(x, skip, wp, scale, shift, head_w, head_b, Co):
    """K4+K5 fused: last decoder stage + 1x1 head + de-slice -> BHWC [B, 16*2H, 2W, Co/16]."""
    for t in (x, skip, wp, scale, shift, head_w, head_b):
        _chk(t, "unet_up_head_fwd")
    B, H, W, C0 = x.shape
    C1 = skip.shape[3] if skip is not None else 0
    if wp.numel() != 16 * Co * (C0 + C1) or head_w.numel() != Co * Co or head_b.numel() != Co or Co not in (16, 32):
        raise RuntimeError("m2h.unet_up_head_fwd: bad weight sizes")
    out = torch.empty((B, 32 * H, 2 * W, Co // 16), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        M = B * H * W
        meta = {"kernel": igemm_config(Co), "M": 4 * M, "N": Co, "K": 4 * (C0 + C1), "flops": 2.0 * 4 * M * Co * (4 * (C0 + C1) + Co),
                "bytes": 4.0 * (x.numel() + (skip.numel() if skip is not None else 0) + out.numel() + wp.numel())}
        _timed("unet_up_head_fwd", meta, x.device,
               lambda: _lib.check(lib.m2h_unet_up_head_fwd(_ptr(x), _ptr(skip), _ptr(wp), _ptr(scale), _ptr(shift), _ptr(head_w), _ptr(head_b),
                                                           _ptr(out), B, H, W, C0, C1, Co, _stream(x)), "m2h_unet_up_head_fwd"))
    return out


def acoustic_mem_small(pred_mono, prev_mem, not_done, w0p, w1p):
    """AcousticMem's forward for a small batch in one launch (m2h_acoustic_mem_small_fwd): BHWC [B,512,32,1] x 2 (+ not-done flags [B])
    -> [B,512,32,1]; w0p / w1p: the packed conv weights [32, 288] / [16, 288]."""
    for t in (pred_mono, prev_mem, not_done, w0p, w1p):
        _chk(t, "acoustic_mem_small")
    B, F, T, C = pred_mono.shape
    if C != 1 or prev_mem.shape != pred_mono.shape or w0p.numel() != 32 * 288 or w1p.numel() != 16 * 288:
        raise RuntimeError("m2h.acoustic_mem_small: expected [B,512,32,1] inputs and [32,288] / [16,288] packed weights")
    if not_done is not None and not_done.numel() != B:
        raise RuntimeError("m2h.acoustic_mem_small: one not-done flag per batch row")
    out = torch.empty_like(pred_mono)
    with torch.cuda.device(pred_mono.device):
        _lib.check(_lib.load().m2h_acoustic_mem_small_fwd(_ptr(pred_mono), _ptr(prev_mem), _ptr(not_done), _ptr(w0p), _ptr(w1p), _ptr(out),
                                                          B, F, T, _stream(pred_mono)), "m2h_acoustic_mem_small_fwd")
    return out
