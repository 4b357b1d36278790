#!/usr/bin/env python3
"""Forward / input-gradient launches of the policy's encoders at the 280-sample update batch under a tuning knob (tuning tool).
usage: python tools/enc_fwd_bench.py [KNOB VALUE]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import functional as MF, ops  # noqa: E402

dev = torch.device("cuda", 0)
if len(sys.argv) > 2:
    ops.debug_set(int(sys.argv[1]), int(sys.argv[2]))
SH = [  # name, B, H, W, Cin, Cout, k, stride, pad
    ("visual.conv0", 280, 128, 128, 4, 32, 8, 4, 0),
    ("visual.conv1", 280, 31, 31, 32, 64, 4, 2, 0),
    ("visual.conv2", 280, 14, 14, 64, 32, 3, 1, 0),
]
SH = [s for s in SH if s]
g = torch.Generator(device=dev).manual_seed(0)
for name, B, H, W, ci, co, k, st, pad in SH:
    x = torch.randn(B, H, W, ci, device=dev, generator=g)
    w = torch.randn(co, ci, k, k, device=dev, generator=g) * 0.05
    wp = ops.pack_conv_weight_ex(w, ci, ci)
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dy = torch.randn(B, Ho, Wo, co, device=dev, generator=g)
    for label, fn in (("fwd", lambda: ops.conv2d_nhwc(x, wp, co, k, k, stride=st, pad=pad, slope=0.0)),
                      ("dgrad", lambda: MF.conv_dgrad(dy, w, (H, W), st, pad))):
        if label == "dgrad" and name == "visual.conv0":
            continue
        for _ in range(3):
            fn()
        kern = ops.last_kernel()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        M, K = B * Ho * Wo, k * k * ci
        print("%-13s %-5s M=%-7d N=%-3d K=%-5d %8.1f us %6.1f TF/s  %s" % (name, label, M, co, K, us, 2.0 * M * co * K / us / 1e6, kern))
