#!/usr/bin/env python3
"""Per-layer A/B micro-benchmark of the igemm conv engine on the GPU (tuning tool, not a test).
usage: python tools/layer_bench.py [--batch 256] [--tm 256] [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import ops  # noqa: E402


GRAPH = False


def time_fn(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if GRAPH:   # small launches are host-bound when enqueued from Python: capture `reps` launches once, time the replay
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tm", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", default="auto;splitk=-1;stages=2;splitk=-1,stages=2")
    ap.add_argument("--graph", action="store_true", help="time a HIP-graph replay of the repetitions (launch-bound shapes)")
    a = ap.parse_args()
    global GRAPH
    GRAPH = a.graph
    dev = torch.device("cuda", 0)
    B, T = a.batch, a.tm
    layers = []
    # encoder (binSep): H=32
    chans = [32, 64, 128, 256, 512, 512]
    H, W = 32, T
    for i in range(5):
        layers.append(("down%d" % i, "down", B, H, W, chans[i], 0, chans[i + 1]))
        H //= 2
        W //= 2
    ups = [(512, 0, 512), (512, 512, 256), (256, 256, 128), (128, 128, 64), (64, 64, 32), (64, 64, 16)]
    H, W = 1, T // 32
    for i, (c0, c1, co) in enumerate(ups):
        if i == 5:
            H, W = 16, T // 2
        layers.append(("up%d(N=%d)" % (min(i, 4), co), "up", B, H, W, c0, c1, co))
        if i < 4:
            H *= 2
            W *= 2
    layers.append(("up4+head(32)", "uphead", B, 16, T // 2, 64, 64, 32))
    layers.append(("up4+head(16)", "uphead", B, 16, T // 2, 64, 64, 16))
    layers.append(("head32", "head", B, 32, T, 32, 0, 32))
    layers.append(("head16", "head", B, 32, T, 16, 0, 16))

    variants = []
    for v in a.variants.split(";"):
        kn = {i: 0 for i in list(range(25)) + [26]}
        if v != "auto":
            for kv in v.split(","):
                k, val = kv.split("=")
                kn[{"splitk": 0, "stages": 1, "wide": 2, "skinny": 3, "n16": 4, "stagger": 5, "pp": 6, "lds": 7, "pmaj": 8, "fast": 9, "math": 14, "tap": 15, "tapbm": 16, "gather": 24, "big": 26}[k]] = int(val)
        variants.append((v, kn))

    print("%-12s %10s %8s %6s | " % ("layer", "M", "K", "N") + " | ".join("%22s" % v for v, _ in variants))
    tot = [0.0] * len(variants)
    for name, kind, B_, H, W, c0, c1, co in layers:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(B_, H, W, c0, device=dev, generator=g)
        sc = torch.rand(co, device=dev, generator=g) + 0.5
        sh = torch.randn(co, device=dev, generator=g) * 0.1
        if kind == "down":
            wp = torch.randn(co, 16 * c0, device=dev, generator=g) * 0.05
            fn = lambda: ops.unet_down_fwd(x, wp, sc, sh, co)
            M, K = B_ * (H // 2) * (W // 2), 16 * c0
        elif kind == "up":
            skip = torch.randn(B_, H, W, c1, device=dev, generator=g) if c1 else None
            wp = torch.randn(4, co, 4 * (c0 + c1), device=dev, generator=g) * 0.05
            fn = lambda: ops.unet_up_fwd(x, skip, wp, sc, sh, co)
            M, K = 4 * B_ * H * W, 4 * (c0 + c1)
        elif kind == "uphead":
            skip = torch.randn(B_, H, W, c1, device=dev, generator=g)
            wp = torch.randn(4, co, 4 * (c0 + c1), device=dev, generator=g) * 0.05
            hw, hb = torch.randn(co, co, device=dev, generator=g) * 0.2, torch.randn(co, device=dev, generator=g) * 0.1
            fn = lambda: ops.unet_up_head_fwd(x, skip, wp, sc, sh, hw, hb, co)
            M, K = 4 * B_ * H * W, 4 * (c0 + c1)
        else:
            wp = torch.randn(co, c0, device=dev, generator=g) * 0.2
            fn = lambda: ops.unet_head_fwd(x, wp, sh, co)
            M, K = B_ * H * W, c0
        flops = 2.0 * M * K * co
        cells = []
        for vi, (v, kn) in enumerate(variants):
            for k, val in kn.items():
                ops.debug_set(k, val)
            us = time_fn(fn, a.reps)
            tot[vi] += us
            cells.append("%9.1f us %6.1f TF/s" % (us, flops / us / 1e6))
        print("%-12s %10d %8d %6d | " % (name, M, K, co) + " | ".join(cells))
    for k in list(range(25)) + [26]:
        ops.debug_set(k, 0)
    print("%-12s %26s | " % ("total(us)", "") + " | ".join("%22.1f" % t for t in tot))


if __name__ == "__main__":
    main()
