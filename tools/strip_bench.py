#!/usr/bin/env python3
"""Micro-benchmark of the strip-walker kernels (csrc/conv_strip.hip) at the benchmark shape, one variant at a time (tuning tool):
first stage unmasked (binSep) / masked (bin2mono), last stage N = 32 / 16; algorithmic bytes and FLOP beside each time.
usage: python tools/strip_bench.py [--batch 256] [--tm 256] [--reps 20] [--only conv1,conv1m,last32,last16]
Under rocprofv3 (--kernel-trace --stats / --pmc ...) each variant is one kernel name, so the per-kernel rows separate them."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import ops, synthetic  # noqa: E402


def time_fn(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
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
    ap.add_argument("--only", default="conv1,conv1m,last32,last16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, T = a.batch, a.tm
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
    pol.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 1).items()})
    pol = pol.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(5)
    mix = torch.log1p(torch.randn(B, 512, T, 2, device=dev, generator=g).abs())
    masks = torch.randn(B, 512, T, 2, device=dev, generator=g) * 0.7 + 0.3
    cls_val = torch.randint(1, 12, (B,), device=dev, generator=g).float()
    only = set(a.only.split(","))
    rows = []
    for name, enc, masked in (("conv1", pol.binSep_enc.passive_sep_encoder, False), ("conv1m", pol.bin2mono_enc.passive_sep_encoder, True)):
        if name not in only:
            continue
        (_wp, scale, shift, table, _co) = enc._packed()[0]
        wreg = ops.pack_strip_conv1(enc.cnn[0][0].weight.detach().contiguous())
        fn = lambda: ops.strip_conv1_fwd(mix, masks if masked else None, wreg, scale, shift, table, cls_val if table is not None else None)  # noqa: E731
        by = 4.0 * B * 512 * T * 2 * (2 if masked else 1) + 4.0 * B * 16 * (T // 2) * 64
        fl = 2.0 * B * 16 * (T // 2) * 64 * 512
        rows.append((name, time_fn(fn, a.reps), by, fl))
    H, W = 16, T // 2
    x = ops.split32(torch.relu(torch.randn(B, H, W, 64, device=dev, generator=g)))
    skip = ops.split32(torch.randn(B, H, W, 64, device=dev, generator=g))
    for name, dec, Co in (("last32", pol.binSep_dec.passive_sep_decoder, 32), ("last16", pol.bin2mono_dec.passive_sep_decoder, 16)):
        if name not in only:
            continue
        ups, (hw, hb, _hco) = dec._packed()
        wp, scale, shift, _co = ups[4]
        wsp = ops.split32(wp)
        fn = lambda: ops.strip_last_fwd(x, skip, wsp, scale, shift, hw, hb, Co)  # noqa: E731
        by = 4.0 * 2 * B * H * W * 64 + 4.0 * B * 512 * T * (Co // 16)
        fl = 2.0 * 4 * B * H * W * Co * (512 + Co)
        rows.append((name, time_fn(fn, a.reps), by, fl))
    for name, us, by, fl in rows:
        print("%-8s %8.1f us   %7.1f MB -> %5.2f TB/s algorithmic (floor %5.1f us at 6.3 TB/s)   %6.1f GFLOP -> %6.1f TFLOP/s (bf16x3 ceiling 833)"
              % (name, us, by / 1e6, by / us / 1e6, by / 6.3e6, fl / 1e9, fl / us / 1e6))


if __name__ == "__main__":
    main()
