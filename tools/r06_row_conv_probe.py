"""update_sep's first conv (3x3, 32 -> 32 or NOUT=16, bf16x3 image-row kernel) alone at 1 680 samples.  The record profiles/r06_row_conv_probe.txt was made with
two TEMPORARY knobs this tool drove (13: resident workgroups, 17: chunk order) -- they are not in the library; without them every argument times the shipped launch."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "move2hear-active-av-separation_amd"))
import torch
from m2h import ops

dev = torch.device("cuda", 0)
ops.set_math_mode(ops.MATH_BF16X3)
B = 1680
x = torch.randn(B, 32, 32, 32, device=dev)
NO = int(os.environ.get("NOUT", "32"))
w = torch.randn(NO, 32, 3, 3, device=dev) * 0.05
wp = ops.pack_conv_weight_ex(w, 32, 32)
for cap in [int(a) for a in sys.argv[1:]] or [0]:
    ops.debug_set(13, abs(cap))
    ops.debug_set(17, -1 if cap < 0 else 0)      # (negative block count: plain chunk order)
    for _ in range(3):
        y = ops.conv2d_nhwc(x, wp, NO, 3, 3, stride=1, pad=1, slope=0.0, name="probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d_nhwc(x, wp, NO, 3, 3, stride=1, pad=1, slope=0.0, name="probe")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("blocks %4d: %.1f us  (%.2f TB/s on x + y = %.0f MB)  %s" % (cap, us, (x.numel() + y.numel()) * 4 / us / 1e6, (x.numel() + y.numel()) * 4 / 1e6, ops.last_kernel()))
