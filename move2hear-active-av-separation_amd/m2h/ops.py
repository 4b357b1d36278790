"""torch-tensor front end of the C-ABI (include/m2h.h).  PyTorch supplies device memory and the current
HIP stream only; all arithmetic happens in libm2h.so.  Every function raises RuntimeError when the library
is missing or a tensor is not a contiguous fp32 CUDA(HIP) tensor -- there is no fallback.
"""
import ctypes

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


def igemm_config(N):
    """Name of the igemm_f32_kernel instantiation conv_igemm.hip picks for N output channels."""
    return "igemm_f32<128,128>" if N > 64 else ("igemm_f32<128,64>" if N > 32 else "igemm_f32<128,32>")


def _timed(name, meta, dev, fn):
    if _timing is None:
        return fn()
    st = torch.cuda.current_stream(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r = fn()
    e1.record(st)
    _timing.append((name, meta, e0, e1))
    return r


def _workspace(nbytes, dev):
    """Split-K scratch from the caching allocator (None when the launch does not split)."""
    if nbytes == 0:
        return None, 0
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    return ws, nbytes


def debug_set(knob, value):
    _lib.check(_lib.load().m2h_debug_set(int(knob), int(value)), "m2h_debug_set")


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
