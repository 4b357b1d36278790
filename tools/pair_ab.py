#!/usr/bin/env python3
"""Interleaved A/B timing of the headline separator pair under different m2h_tuning_set knob settings (tuning tool, not a test).

Two bench.py runs land on different boxes and clock states (+-5 %); here every variant's HIP graph is captured once and the
variants are replayed round-robin, so a difference of a per cent or two between them is visible.
usage: python tools/pair_ab.py --variants "16=256;auto" [--rounds 8] [--steps 10]      (variant = knob=value[,knob=value...])"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from m2h import ops  # noqa: E402
from m2h.graphs import GraphedSeparatorPair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tm", type=int, default=256)
    ap.add_argument("--layers", action="store_true", help="also time every kernel of both U-Nets per variant (HIP events, no graph)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pol, _sd = bench.make_policy(dev)
    mix, tc = bench.make_inputs(dev, a.batch, a.tm, 1000)
    obs = {"mixed_bin_audio_mag": mix, "target_class": tc}
    variants = []
    with ops.math_scope(ops.MATH_BF16X3):
        for v in a.variants.split(";"):
            kn = {}
            if v != "auto":
                for kv in v.split(","):
                    k, val = kv.split("=")
                    kn[int(k)] = int(val)
            for k, val in kn.items():
                ops.debug_set(k, val)
            g = GraphedSeparatorPair(pol, obs)   # the knobs are read at launch time = at capture
            g()
            torch.cuda.synchronize()
            for k in kn:
                ops.debug_set(k, 0)
            variants.append((v, g, []))
        for _ in range(a.rounds):
            for v, g, ts in variants:
                g()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    g()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.steps)
    if a.layers:
        from m2h.rl.models.separator_cnn import unet_forward
        names = ["slice"] + ["down%d" % i for i in range(5)] + ["up%d" % i for i in range(5)]
        rows = {}
        with ops.math_scope(ops.MATH_BF16X3), torch.no_grad():
            for v, _g, _ts in variants:
                kn = {} if v == "auto" else {int(kv.split("=")[0]): int(kv.split("=")[1]) for kv in v.split(",")}
                for k, val in kn.items():
                    ops.debug_set(k, val)
                acc = [[0.0] * 11, [0.0] * 11]
                reps = 6
                for rep in range(reps + 1):
                    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(12)] for _ in range(2)]
                    for ev in evs:
                        for e in ev:
                            e.record()
                    masks = unet_forward(pol.binSep_enc.passive_sep_encoder, pol.binSep_dec.passive_sep_decoder, mix, None, tc, events=evs[0])
                    unet_forward(pol.bin2mono_enc.passive_sep_encoder, pol.bin2mono_dec.passive_sep_decoder, mix, masks, events=evs[1])
                    torch.cuda.synchronize()
                    if rep:
                        for u in range(2):
                            for i in range(11):
                                acc[u][i] += evs[u][i].elapsed_time(evs[u][i + 1]) * 1e3 / reps
                for k in kn:
                    ops.debug_set(k, 0)
                rows[v] = acc
        print("%-8s " % "layer" + " ".join("%22s" % v for v, _g, _t in variants) + "   (us: binSep / bin2mono)")
        for i in range(11):
            print("%-8s " % names[i] + " ".join("%10.1f /%10.1f" % (rows[v][0][i], rows[v][1][i]) for v, _g, _t in variants))
    for v, _g, ts in variants:
        ts = sorted(ts)
        print("%-24s median %.4f ms  min %.4f  max %.4f   (%d rounds x %d steps)" % (v, ts[len(ts) // 2], ts[0], ts[-1], a.rounds, a.steps))


if __name__ == "__main__":
    main()
