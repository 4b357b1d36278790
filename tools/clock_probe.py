#!/usr/bin/env python3
"""Diagnostic (not product): the shader clock the chip holds INSIDE the headline's conv engines, from the two s_memtime / s_memrealtime
stamps the M2H_CLOCK_DIAG build puts around each block's k-loop (MI355X_MICROARCH.md, DVFS give-back item 6: delta s_memtime / delta
s_memrealtime x 100 MHz, after ~1 s of back-to-back launches on random data, median over workgroups).  Runs the four shared-patch layer
shapes of the benchmark batch (B 256 x 512 x 256: second / third encoder stage, third / fourth decoder stage) on the diagnostic copy of the
library (m2h/_lib.py build_clock_diag) and prints ONE JSON line.  bench.py runs this as a child process after its timed regions and copies
the dominant instantiation's clock into roofline.clock_ghz: a 9 % box-to-box spread of ms_per_step is then told from a code change.
usage: python3 tools/clock_probe.py [--seconds 1.0]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=1.0)
args = ap.parse_args()
if not os.path.exists(_lib.CLOCK_DIAG_LIB):
    print(json.dumps({"error": "no diagnostic library (m2h._lib.build_clock_diag)"}))
    sys.exit(0)
_lib.LIB_PATH = _lib.CLOCK_DIAG_LIB
from m2h import ops  # noqa: E402

lib = _lib.load()
reader = lib.m2h_diag_read_clocks_patch
reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda", 0)
ops.set_math_mode(ops.MATH_BF16X3)
fmt = ops.FMT_SRC_SPLIT | ops.FMT_W_SPLIT | ops.FMT_DST_SPLIT


def conv_t_layer(x, x2, wp, Co, sc, sh):
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


# (B, H, W, Ci, Co, stage of the headline pair, transposed): the shared-patch engine's layers at the benchmark batch
CASES = [(256, 8, 64, 128, 256, "down3 (conv 128->256)", False), (256, 16, 128, 64, 128, "down2 (conv 64->128)", False),
         (256, 8, 64, 128, 64, "up4 (convT 256->64, 512x64 tiles)", True), (256, 4, 32, 256, 128, "up3 (convT 512->128)", True)]
out = {}
for (B, H, W, Ci, Co, label, transposed) in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops.split32(torch.randn(B, H, W, Ci, device=dev, generator=g))
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    if transposed:
        x2 = ops.split32(torch.randn(B, H, W, Ci, device=dev, generator=g))
        wp = ops.split32(ops.pack_convT_weight(torch.randn(2 * Ci, Co, 4, 4, device=dev, generator=g) * 0.05))
        nk = 4 * 2 * Ci // 32
        run = lambda: conv_t_layer(x, x2, wp, Co, sc, sh)  # noqa: E731
    else:
        wp = ops.split32(torch.randn(Co, 16 * Ci, device=dev, generator=g) * 0.05)
        nk = 16 * Ci // 32
        run = lambda: ops.conv2d_nhwc(x, wp, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2, operand_format=fmt)  # noqa: E731
    t0 = time.time()
    while time.time() - t0 < args.seconds:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    kernel = ops.last_kernel()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros((2048, 16), np.uint64)
    reader(buf.ctypes.data, 2048)
    b = buf[buf[:, 1] > 0]
    clk = b[:, 0].astype(np.float64) / b[:, 1].astype(np.float64) * 0.1
    tiles = np.maximum(b[:, 6].astype(np.float64), 1.0)
    out[label] = {"kernel": kernel, "clock_ghz": round(float(np.median(clk)), 3), "clock_ghz_min": round(float(clk.min()), 3), "clock_ghz_max": round(float(clk.max()), 3),
                  "workgroups_stamped": int(len(b)), "us_per_launch": round(e0.elapsed_time(e1) / 20 * 1e3, 1),
                  "cycles_per_k_tile": round(float(np.median(b[:, 0] / tiles)) / nk, 0)}
print(json.dumps({"what": "in-kernel shader clock (delta s_memtime / delta s_memrealtime x 100 MHz around the k-loop, median over workgroups) of the "
                          "shared-patch engine's layers at the benchmark batch, diagnostic build, ~%.1f s of back-to-back launches each" % args.seconds,
                  "layers": out}))
