#!/usr/bin/env python3
"""Tuning probe: the separator pair over a 256-batch as ONE stream vs TWO concurrent half-batch streams (kernel tails and
kernel-boundary drains of one half overlapping the other's body)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from m2h import ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pol, _ = bench.make_policy(dev)
    B, T = 256, 256
    mix, tc = bench.make_inputs(dev, B, T, 1000)
    ops.set_math_mode(ops.MATH_BF16X3)

    def pair(m, c):
        with torch.no_grad():
            masks = pol.get_binSepMasks({"mixed_bin_audio_mag": m, "target_class": c})
            return masks, pol.convert_bin2mono(masks, mixed_audio=m)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    print("one stream  B=256: %.3f ms" % timeit(lambda: pair(mix, tc)))
    for parts in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        chunks = [(mix[i * B // parts:(i + 1) * B // parts], tc[i * B // parts:(i + 1) * B // parts]) for i in range(parts)]

        def multi():
            cur = torch.cuda.current_stream()
            for s, (m, c) in zip(streams, chunks):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    pair(m, c)
            for s in streams:
                cur.wait_stream(s)
        print("%d streams x B=%d: %.3f ms" % (parts, B // parts, timeit(multi)))
        g = torch.cuda.CUDAGraph()
        multi()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            multi()
        print("%d streams x B=%d, graph: %.3f ms" % (parts, B // parts, timeit(g.replay)))
    ops.set_math_mode(ops.MATH_FP32)


if __name__ == "__main__":
    main()
