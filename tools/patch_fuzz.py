#!/usr/bin/env python3
"""Randomised cross-check of the shared-patch engine (csrc/conv_patch.hip) against the engines it replaces: random batch, power-of-two
pixel grids, channel counts (multiples of 32), one / two sources, conv / transposed conv, both patch forms (knob 36 = 2 / 3), each
layer run through m2h_conv_igemm_f32 with the engine forced and with it switched off (knob 36 = -1: LDS-DMA / register engines),
compared element by element.  usage: python tools/patch_fuzz.py [--cases 200] [--seed 0]"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

from m2h import ops  # noqa: E402
from test_gpu_patch import _layer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(a.seed)
    ops.set_math_mode(ops.MATH_BF16X3)
    ran = skipped = 0
    worst = 0.0
    for case in range(a.cases):
        transposed = rnd.random() < 0.5
        Hq, Wq = 2 ** rnd.randint(0, 4), 2 ** rnd.randint(1, 6)
        H, W = (Hq, Wq) if transposed else (2 * Hq, 2 * Wq)
        B = rnd.randint(1, 9)
        C0 = 32 * rnd.randint(1, 6)
        C1 = 32 * rnd.randint(1, 4) if (transposed and rnd.random() < 0.6) else 0
        Co = 64 * rnd.randint(1, 5)
        if B * Hq * Wq <= 64 or B * H * W * max(C0, C1) > 64 * 1024 * 1024:
            skipped += 1
            continue
        x = ops.split32(torch.randn(B, H, W, C0, device=dev, generator=g))
        x2 = ops.split32(torch.randn(B, H, W, C1, device=dev, generator=g)) if C1 else None
        Ci = C0 + C1
        if transposed:
            wp = ops.split32(ops.pack_convT_weight(torch.randn(Ci, Co, 4, 4, device=dev, generator=g) * (1.0 / (4 * Ci) ** 0.5)))
        else:
            wp = ops.split32(ops.pack_conv_weight(torch.randn(Co, Ci, 4, 4, device=dev, generator=g) * (1.0 / (16 * Ci) ** 0.5)))
        scale = torch.rand(Co, device=dev, generator=g) + 0.5
        shift = torch.randn(Co, device=dev, generator=g) * 0.1
        args = (x, x2, wp, Co, transposed, scale, shift, 0.0 if transposed else 0.2)
        knob = rnd.choice((2, 3))
        grid = rnd.choice((0, 0, 8, 16, 24, 40))   # workgroups of the engine's persistent launch (0: one per CU): other tile sequences per workgroup
        try:
            ops.debug_set(36, knob)
            ops.debug_set(10, grid)
            got, label = _layer(*args)
            ops.debug_set(36, -1)
            ref, ref_label = _layer(*args)
        finally:
            ops.debug_set(36, 0)
            ops.debug_set(10, 0)
        if not label.startswith("igemm_patch"):   # a shape the engine refuses (tap window, tiny images): nothing to compare
            skipped += 1
            continue
        ran += 1
        err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
        worst = max(worst, err)
        ok = err < 5e-5 and bool(torch.isfinite(got).all())
        if not ok or case % 20 == 0:
            print("%s case %3d %-5s B=%d grid %dx%d C0=%d C1=%d Co=%d knob=%d %s vs %s: max err %.2e" % (
                "ok  " if ok else "FAIL", case, "convT" if transposed else "conv", B, Hq, Wq, C0, C1, Co, knob, label, ref_label, err))
        if not ok:
            sys.exit(1)
    print("cases run %d, skipped %d, worst max-error %.2e of the largest output" % (ran, skipped, worst))


if __name__ == "__main__":
    main()
