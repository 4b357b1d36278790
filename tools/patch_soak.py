#!/usr/bin/env python3
"""Race screen for the shared-patch engine's counted waits (tuning tool, not a test): the benchmark pair is replayed N times from
its HIP graph and every replay's outputs must equal the first replay's bit for bit (an LDS-DMA read placed one wait too early
passes reference checks whenever the DMA happens to land first: rare wrong tiles that come and go -- cdna_hip_programming.md,
"Read a staged buffer one phase AFTER the wait that retires it").  Also at two other batch sizes (ragged last tiles, other tiles per CU).
usage: python tools/patch_soak.py [--replays 300]"""
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
    ap.add_argument("--replays", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pol, _sd = bench.make_policy(dev)
    bad = 0
    with ops.math_scope(ops.MATH_BF16X3):
        for batch, tm in ((256, 256), (200, 256), (77, 128)):
            mix, tc = bench.make_inputs(dev, batch, tm, 1000 + batch)
            g = GraphedSeparatorPair(pol, {"mixed_bin_audio_mag": mix, "target_class": tc})
            m0, o0 = (t.clone() for t in g())
            torch.cuda.synchronize()
            n = a.replays if batch == 256 else a.replays // 3
            mism = 0
            for i in range(n):
                m, o = g()
                if i % 8 == 7 or i == n - 1:     # compare every eighth replay (the compare itself is a device reduction + one host read)
                    if not (torch.equal(m, m0) and torch.equal(o, o0)):
                        mism += 1
            torch.cuda.synchronize()
            print("batch %d x %d frames: %d replays, %d compared sets differ from the first replay" % (batch, tm, n, mism))
            bad += mism
    print("OK" if bad == 0 else "MISMATCH")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
