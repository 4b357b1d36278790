#!/usr/bin/env python3
"""A/B of the bf16x3 conv engine with on-the-fly operand split vs operands already in the split32 layout (tuning tool)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402
from m2h import ops  # noqa: E402


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    B = 256
    ops.set_math_mode(ops.MATH_BF16X3)
    for name, H, W, Ci, Co in (("down0", 32, 256, 32, 64), ("down1", 16, 128, 64, 128), ("down2", 8, 64, 128, 256), ("down3", 4, 32, 256, 512)):
        x = torch.randn(B, H, W, Ci, device=dev, generator=g)
        wp = torch.randn(Co, 16 * Ci, device=dev, generator=g) * 0.05
        sc, sh = torch.rand(Co, device=dev, generator=g) + 0.5, torch.randn(Co, device=dev, generator=g) * 0.1
        out = torch.empty(B, H // 2, W // 2, Co, device=dev)
        xs, ws = ops.split32(x), ops.split32(wp)
        a = t(lambda: ops.conv2d_nhwc(x, wp, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2, out=out))
        b = t(lambda: ops.conv2d_nhwc(xs, ws, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2, out=out, operand_format=3))
        c = t(lambda: ops.conv2d_nhwc(xs, ws, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2, out=out, operand_format=7))
        fl = 2.0 * B * (H // 2) * (W // 2) * Co * 16 * Ci
        print("%-6s on-the-fly %7.1f us %6.1f TF | presplit %7.1f us %6.1f TF | presplit+split out %7.1f us" % (name, a, fl / a / 1e6, b, fl / b / 1e6, c))
    ops.set_math_mode(ops.MATH_FP32)


if __name__ == "__main__":
    main()
