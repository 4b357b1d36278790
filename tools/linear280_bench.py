#!/usr/bin/env python3
"""The policy's dense layers at the 280-row update batch (nn.Linear forward / input gradient as 1x1 convs on the igemm engine) under a
forced split-K factor (tuning tool).   usage: python tools/linear280_bench.py [S ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import ops  # noqa: E402

dev = torch.device("cuda", 0)
SH = [("visual.fc", 280, 512, 4608), ("fc.dgrad", 280, 4608, 512), ("gru.ih", 280, 1536, 1536), ("audio.fc", 280, 512, 32), ("M=14 gru.ih", 14, 1536, 1536)]
g = torch.Generator(device=dev).manual_seed(0)
G = int(os.environ.get("GATHER", "0"))
ops.debug_set(24, G)
for S in [int(a) for a in sys.argv[1:]] or [0]:
    ops.debug_set(0, S)
    for name, M, N, K in SH:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.02
        fn = lambda: ops.linear(x, w, None)
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
        print("S=%-3d %-12s M=%-4d N=%-5d K=%-5d %8.1f us %6.1f TF/s  %s" % (S, name, M, N, K, us, 2.0 * M * N * K / us / 1e6, kern))
ops.debug_set(0, 0)
ops.debug_set(24, 0)
