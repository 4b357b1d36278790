#!/usr/bin/env python3
"""Does an update_sep epoch get cheaper when the 1 680 stored samples go through AcousticMem's five training kernels in chunks whose
activations stay in the memory-side cache?  Times forward + loss + backward of the whole batch against the same work in 2 / 4 / 8 / 16
chunks (same kernels, bf16x3 math, one HIP graph per chunk count).  usage: python tools/r05_sep_chunk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import functional as MF  # noqa: E402
from m2h import ops  # noqa: E402
from m2h.rl.models.memory_nets import AcousticMem  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    B = 1680
    mem = AcousticMem(use_ddppo=True).to(dev).train()
    pred = torch.rand(B, 512, 32, 1, device=dev)
    prev = torch.rand(B, 512, 32, 1, device=dev)
    masks = torch.ones(B, 1, device=dev)
    gt = torch.rand(B, 512, 32, 2, device=dev)
    unit = MF.unit_grad(dev)
    with ops.math_scope(ops.MATH_BF16X3):
        sliced = mem.slice_inputs(pred, prev, masks)

        def epoch(nch):
            n = B // nch
            for c in range(nch):
                s = slice(c * n, (c + 1) * n)
                loss = mem.l1_loss_masked(pred[s], prev[s], masks[s], gt[s], 0, sliced=sliced[s])
                loss.backward(unit)

        from m2h import graphs
        for nch in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,4,8,16,1,8,16'.split(','))]:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    epoch(nch)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with graphs.capture(g):          # one HIP graph per chunk count: no host time between the launches
                epoch(nch)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            print("chunks %2d (%4d samples each): %.1f us per epoch (graph replay, %d kernels)" % (nch, B // nch, e0.elapsed_time(e1) * 100.0, g._m2h_kernels), flush=True)


if __name__ == "__main__":
    main()
