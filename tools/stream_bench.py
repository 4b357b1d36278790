"""Per-layer time of the rollout batch's weight-streaming layers: the stream split-K kernel (csrc/conv_stream.hip) against the 16-row
kernels it replaces (knob 5 = -1), and its block-count target (knob 6).  Each layer is timed as a HIP graph of 20 launches on 20
DIFFERENT weight copies (a layer's weights are read once per rollout step, between ~150 MB of other traffic: never from a warm L2).
usage: gpurun -- python3 tools/stream_bench.py [--targets 128,256,512]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
from m2h import ops  # noqa: E402


def layers(dev, copies):
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    out = []
    for name, (B, H, W, Ci, Co) in (("down3 14x8x8x128->256", (14, 8, 8, 128, 256)), ("down4 14x4x4x256->512", (14, 4, 4, 256, 512)),
                                    ("down5 14x2x2x512->512", (14, 2, 2, 512, 512))):
        x = r(B, H, W, Ci)
        wps = [r(Co, 16 * Ci) * 0.02 for _ in range(copies)]
        sc, sh = r(Co), r(Co)
        live = 4 if H == 2 else 16
        out.append((name, Co * live * Ci * 4, [lambda wp=wp, x=x, sc=sc, sh=sh, Co=Co: ops.unet_down_fwd(x, wp, sc, sh, Co) for wp in wps]))
    for name, (B, H, W, C0, C1, Co) in (("up1 14x1x1x512->512", (14, 1, 1, 512, 0, 512)), ("up2 14x2x2x(512+512)->256", (14, 2, 2, 512, 512, 256)),
                                        ("up3 14x4x4x(256+256)->128", (14, 4, 4, 256, 256, 128))):
        x = r(B, H, W, C0)
        skip = r(B, H, W, C1) if C1 else None
        wps = [r(4, Co, 4 * (C0 + C1)) * 0.02 for _ in range(copies)]
        sc, sh = r(Co), r(Co)
        live = 4 if H == 1 else 16
        out.append((name, Co * live * (C0 + C1) * 4, [lambda wp=wp, x=x, skip=skip, sc=sc, sh=sh, Co=Co: ops.unet_up_fwd(x, skip, wp, sc, sh, Co) for wp in wps]))
    x = r(14, 4608)
    ws = [r(512, 4608) * 0.02 for _ in range(copies)]
    b = r(512)
    out.append(("visual fc 14x4608->512", 512 * 4608 * 4, [lambda w=w, x=x, b=b: ops.linear(x, w, b, slope=0.0) for w in ws]))
    x = r(14, 1536)
    ws = [r(1536, 1536) * 0.02 for _ in range(copies)]
    out.append(("linear 14x1536->1536", 1536 * 1536 * 4, [lambda w=w, x=x: ops.linear(x, w) for w in ws]))
    return out


def time_graph(fns, dev, reps=20):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for f in fns[:2]:
            f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
    label = ops.last_kernel()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns)), label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", default="128,256,512")
    ap.add_argument("--copies", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    print("%-30s %10s | %-34s | %s" % ("layer", "live MB", "16-row kernels (knob 5 = -1)", "stream split-K at block targets " + a.targets))
    for name, nbytes, fns in layers(dev, a.copies):
        ops.debug_set(5, -1)
        t_old, lab_old = time_graph(fns, dev)
        ops.debug_set(5, 0)
        cells = []
        for tgt in [int(t) for t in a.targets.split(",")]:
            ops.debug_set(6, tgt)
            t, lab = time_graph(fns, dev)
            cells.append("%d: %5.1f us %5.2f TB/s%s" % (tgt, t, nbytes / t / 1e6, "" if "stream" in lab else " (" + lab[-14:] + ")"))
        ops.debug_set(6, 0)
        print("%-30s %10.2f | %6.1f us %5.2f TB/s  %-14s | %s" % (name, nbytes / 1e6, t_old, nbytes / t_old / 1e6, lab_old[-14:], "   ".join(cells)))


if __name__ == "__main__":
    main()
