#!/usr/bin/env python3
"""Diagnostic (not product): builds a copy of libm2h with -DM2H_CLOCK_DIAG, runs one split32 U-Net layer on the LDS-DMA engine
back to back for ~2 s and prints the shader clock the chip holds inside its k-loop (delta s_memtime / delta s_memrealtime x
100 MHz) and the cycles per k-tile.  usage: python tools/clock_diag_dma.py [knob27 values, e.g. 0 5 4]
       python tools/clock_diag_dma.py patch [knob36 values, e.g. 0 5 4]     (the shared-patch engine, csrc/conv_patch.hip)"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import _lib  # noqa: E402

diag = "/tmp/libm2h_diag.so"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DM2H_CLOCK_DIAG", "-I" + _lib.INCLUDE, "-I" + _lib.CSRC]
cmd += [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES] + ["-o", diag]
subprocess.check_call(cmd)
_lib.LIB_PATH = diag
from m2h import ops  # noqa: E402

lib = _lib.load()
argv = sys.argv[1:]
patch = bool(argv) and argv[0] == "patch"
if patch:
    argv = argv[1:]
reader = lib.m2h_diag_read_clocks_patch if patch else lib.m2h_diag_read_clocks_dma
reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
KNOB = 36 if patch else 27
dev = torch.device("cuda", 0)
knobs = [int(v) for v in argv] or [0]
ops.set_math_mode(ops.MATH_BF16X3)
fmt = ops.FMT_SRC_SPLIT | ops.FMT_W_SPLIT | ops.FMT_DST_SPLIT
def conv_t_layer(x, x2, wp, Co, sc, sh):
    """One transposed-conv layer (two split32 sources) through m2h_conv_igemm_f32 (as tests/test_gpu_patch.py)."""
    B, H, W, C0 = x.shape
    out = torch.empty((B, 2 * H, 2 * W, Co), device=x.device, dtype=torch.float32)
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), x2.data_ptr(), C0, x2.shape[3]
    a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, H, W
    a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = 1, 2, 2, 0, 0, 0, 0
    a.conv_transpose, a.os = 1, 2
    a.wp, a.N = wp.data_ptr(), Co
    a.scale, a.shift, a.slope, a.cls_table, a.cls_val = sc.data_ptr(), sh.data_ptr(), 0.0, None, None
    a.dst, a.Ho, a.Wo, a.ph, a.pw, a.ldc, a.out_mode = out.data_ptr(), 2 * H, 2 * W, 0, 0, Co, ops.OUT_NHWC
    a.operand_format = fmt
    a.workspace, a.workspace_bytes = None, 0
    with torch.cuda.device(x.device):
        _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), ops._stream(x)), "m2h_conv_igemm_f32")
    return out


CASES = [(256, 8, 64, 128, 256, "down2 K=2048 N=256", False), (256, 16, 128, 64, 128, "down1 K=1024 N=128", False)]
if patch:
    CASES += [(256, 8, 64, 128, 64, "up3 K=4x256 N=64 (512x64 tiles)", True), (256, 4, 32, 256, 128, "up2 K=4x512 N=128", True)]
for (B, H, W, Ci, Co, label, transposed) in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops.split32(torch.randn(B, H, W, Ci, device=dev, generator=g))
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    if transposed:
        x2 = ops.split32(torch.randn(B, H, W, Ci, device=dev, generator=g))
        wp = ops.split32(ops.pack_convT_weight(torch.randn(2 * Ci, Co, 4, 4, device=dev, generator=g) * 0.05))
        nk = 4 * 2 * Ci // 32
        run = lambda: conv_t_layer(x, x2, wp, Co, sc, sh)  # noqa: E731
    else:
        wp = ops.split32(torch.randn(Co, 16 * Ci, device=dev, generator=g) * 0.05)
        nk = 16 * Ci // 32
        run = lambda: ops.conv2d_nhwc(x, wp, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2, operand_format=fmt)  # noqa: E731
    for kv in knobs:
        ops.debug_set(KNOB, kv)
        if not patch:
            ops.debug_set(36, -1)
        t0 = time.time()
        n = 0
        while time.time() - t0 < 2.0:
            for _ in range(50):
                run()
            torch.cuda.synchronize()
            n += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nb = 2048
        buf = np.zeros((nb, 16 if patch else 8), np.uint64)
        reader(buf.ctypes.data, nb)
        b = buf[buf[:, 1] > 0]
        tiles = np.maximum(b[:, 6].astype(np.float64), 1.0) if patch else 1.0
        clk = b[:, 0].astype(np.float64) / b[:, 1].astype(np.float64) * 0.1
        print("%s knob=%d: %.1f us/launch, %d blocks stamped; in-kernel clock median %.3f GHz (min %.3f max %.3f); k-loop cycles median %.0f = %.0f per k-tile (epilogues and tile boundaries included)"
              % (label, kv, us, len(b), np.median(clk), clk.min(), clk.max(), np.median(b[:, 0]), np.median(b[:, 0] / tiles) / nk))
        f = b.astype(np.float64)
        print("    per workgroup (us, median): setup + ring fill %.2f, k-loop %.2f, epilogue %.2f; first start -> last end %.1f"
              % (np.median(f[:, 3] - f[:, 2]) / 100, np.median(f[:, 4] - f[:, 3]) / 100, np.median(f[:, 5] - f[:, 4]) / 100, (f[:, 5].max() - f[:, 2].min()) / 100))
        if patch and f.shape[1] >= 15 and (f[:, 8] > 0).any():
            t = f[f[:, 8] > 0]
            d = lambda a, c: np.median(t[:, c] - t[:, a])  # noqa: E731
            print("    first tile boundary (shader cycles, median over %d workgroups of >= 2 tiles): store_tile %.0f, gap %.0f, then the four k-tiles %.0f %.0f %.0f %.0f"
                  % (len(t), d(8, 9), d(9, 10), d(10, 11), d(11, 12), d(12, 13), d(13, 14)))
    ops.debug_set(KNOB, 0)
    ops.debug_set(36, 0)
