"""AcousticMem's update_sep epoch (forward + backward over the 1680 stored samples) in the two arithmetic modes: values and time.
    python tools/amem_bf16x3.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import ops  # noqa: E402
from m2h.rl.models.memory_nets import AcousticMem  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1680
torch.manual_seed(0)
mem = AcousticMem(use_ddppo=True).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
mono = torch.rand(B, 512, 32, 1, device=dev, generator=g) * 2
prev = torch.rand(B, 512, 32, 1, device=dev, generator=g) * 2
nd = (torch.rand(B, 1, device=dev, generator=g) > 0.3).float()
gy = torch.randn(B, 512, 32, 1, device=dev, generator=g)
with torch.no_grad():
    x = mem.slice_inputs(mono, prev, nd)


def rel(a, b):
    return float((a - b).abs().sum() / b.abs().sum())


def run(mode):
    with ops.math_scope(mode):
        for p in mem.parameters():
            p.grad = None
        out = mem.forward_masked(mono, prev, nd, sliced=x)
        k_fwd = ops.last_kernel()
        out.backward(gy)
        return out.detach(), [p.grad.clone() for p in mem.parameters()], k_fwd


def timeit(mode, reps=10):
    with ops.math_scope(mode):
        for _ in range(2):
            run(mode)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(reps):
            for p in mem.parameters():
                p.grad = None
            e[0].record()
            out = mem.forward_masked(mono, prev, nd, sliced=x)
            e[1].record()
            out.backward(gy)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        return 1e3 * tf / reps, 1e3 * tb / reps


o32, g32, k32 = run(ops.MATH_FP32)
o16, g16, k16 = run(ops.MATH_BF16X3)
print("B=%d  forward kernels: fp32 %r, bf16x3 %r" % (B, k32, k16))
print("rel-L1 bf16x3 vs fp32: out %.3g, dW0 %.3g, dW1 %.3g" % (rel(o16, o32), rel(g16[0], g32[0]), rel(g16[1], g32[1])))
if B <= 128:
    ref = torch.nn.Sequential(torch.nn.Conv2d(32, 32, 3, padding=1, bias=False), torch.nn.ReLU(), torch.nn.Conv2d(32, 16, 3, padding=1, bias=False)).double()
    ref[0].weight.data.copy_(mem.cnn[0].weight.detach().cpu())
    ref[2].weight.data.copy_(mem.cnn[2].weight.detach().cpu())
    y = ref(x.cpu().double().permute(0, 3, 1, 2))                      # [B,16,32,32] -> BHWC [B,512,32,1]
    y = y.reshape(B, 16 * 32, 32, 1)
    y.backward(gy.cpu().double())
    for name, o, gr in (("fp32", o32, g32), ("bf16x3", o16, g16)):
        print("  %-7s vs float64 torch: out %.3g, dW0 %.3g, dW1 %.3g" % (name, rel(o.cpu().double(), y.detach()), rel(gr[0].cpu().double(), ref[0].weight.grad),
                                                                         rel(gr[1].cpu().double(), ref[2].weight.grad)))
f32, b32 = timeit(ops.MATH_FP32)
f16, b16 = timeit(ops.MATH_BF16X3)
print("time per pass (eager, device events): fp32 forward %.0f us backward %.0f us | bf16x3 forward %.0f us backward %.0f us" % (f32, b32, f16, b16))
