#!/usr/bin/env python3
"""Diagnostic (not product): builds a copy of libm2h with -DM2H_CLOCK_DIAG, runs one U-Net layer back to back for ~2 s and
prints the shader clock the chip holds inside the igemm k-loop (delta s_memtime / delta s_memrealtime x 100 MHz)."""
import ctypes
import os
import subprocess
import sys

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
lib.m2h_diag_read_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda", 0)
for (B, H, W, C0, C1, Co, label) in [(256, 2, 16, 512, 512, 256, "up1 K=4096 N=256"), (256, 8, 64, 128, 128, 64, "up3 K=1024 N=64")]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, W, C0, device=dev, generator=g)
    sk = torch.randn(B, H, W, C1, device=dev, generator=g)
    wp = torch.randn(4, Co, 4 * (C0 + C1), device=dev, generator=g) * 0.05
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    import time
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.0:
        for _ in range(50):
            ops.unet_up_fwd(x, sk, wp, sc, sh, Co)
        torch.cuda.synchronize()
        n += 50
    nb = 4 * ((B * H * W + 127) // 128 + 7) // 8 * 8 * ((Co + 127) // 128 if Co > 64 else 1)
    nb = min(nb, 8192)
    buf = np.zeros((nb, 6), np.uint64)
    rc = lib.m2h_diag_read_clocks(buf.ctypes.data, nb)
    ok = buf[:, 1] > 0
    b = buf[ok]
    clk = b[:, 0].astype(np.float64) / b[:, 1].astype(np.float64) * 100e6 / 1e9
    print("%s: %d launches, %d blocks stamped; in-kernel clock median %.3f GHz (min %.3f max %.3f); k-loop cycles median %.0f" %
          (label, n, len(b), np.median(clk), clk.min(), clk.max(), np.median(b[:, 0])))
    # concurrency per CU: HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... ; xcc id separate
    t0r = b[:, 2].min()
    start = (b[:, 2] - t0r).astype(np.float64) / 100.0  # us
    end = (b[:, 3] - t0r).astype(np.float64) / 100.0
    cu = ((b[:, 5] & 0xF) << 16) | (b[:, 4] & 0xFF00)
    ncu = len(np.unique(cu))
    overl = 0
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        iv = sorted((start[i], end[i]) for i in idx)
        for a, bb in zip(iv[:-1], iv[1:]):
            if bb[0] < a[1] - 1.0:
                overl += 1
    print("   k-loop start percentiles (us):", np.percentile(start, [0, 10, 50, 90, 100]).round(1), " end:", np.percentile(end, [0, 10, 50, 90, 100]).round(1))
    dur = end - start
    xcc = (b[:, 5] & 0xF).astype(int)
    print("   duration by XCC:", {int(x): round(float(dur[xcc == x].mean()), 1) for x in np.unique(xcc)})
    bi = np.where(ok)[0]
    gx = nb // 4
    print("   duration by phase:", {int(z): round(float(dur[(bi // gx) == z].mean()), 1) for z in range(4)})
    se = ((b[:, 4] >> 13) & 7).astype(int)
    print("   duration by SE id:", {int(x): round(float(dur[se == x].mean()), 1) for x in np.unique(se)})
    cuid = ((b[:, 4] >> 8) & 15).astype(int)
    print("   duration by CU id:", {int(x): round(float(dur[cuid == x].mean()), 1) for x in np.unique(cuid)})
    clk_b = b[:, 0].astype(np.float64) / b[:, 1].astype(np.float64) * 0.1
    print("   clock by XCC:", {int(x): round(float(clk_b[xcc == x].mean()), 3) for x in np.unique(xcc)})
    print("   distinct CUs %d, blocks/CU %.2f, overlapping block pairs on a CU %d; k-loop span: first start %.1f us, last end %.1f us, median duration %.1f us"
          % (ncu, len(b) / max(ncu, 1), overl, start.min(), end.max(), np.median(end - start)))
