#!/usr/bin/env python3
"""Diagnostic (not product): builds a copy of libm2h with -DM2H_CLOCK_DIAG and prints where a workgroup of the four-phase
transposed-conv kernel (csrc/convt_quad.hip) spends its time: real-time stamps (100 MHz) at start, after the prologue's DMA
issue, after its wait, after the k-loop and after each phase's epilogue; the headline batch of one U-Net through the runner."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import _lib  # noqa: E402

diag = "/tmp/libm2h_diag.so"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DM2H_CLOCK_DIAG", "-DM2H_QUAD_DBG=%d" % int(os.environ.get("M2H_QUAD_DBG", "0")),
       "-I" + _lib.INCLUDE, "-I" + _lib.CSRC]
cmd += [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES] + ["-o", diag]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = diag
import bench  # noqa: E402
from m2h import ops  # noqa: E402

lib = _lib.load()
lib.m2h_diag_read_clocks_quad.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda", 0)
pol, _sd = bench.make_policy(dev)
mix, tc = bench.make_inputs(dev, 256, 256, 1000)
obs = {"mixed_bin_audio_mag": mix, "target_class": tc}
ops.set_math_mode(ops.MATH_BF16X3)
with torch.no_grad():
    for _ in range(3):
        pol.get_binSepMasks(obs)      # the last four-phase launch of this call is the 32-wide last stage (+ head)
    torch.cuda.synchronize()
nb = 2048
buf = np.zeros((nb, 8), np.uint64)
lib.m2h_diag_read_clocks_quad(buf.ctypes.data, nb)
b = buf[buf[:, 7] > 0].astype(np.float64)
t0 = b[:, 0].min()
d = (b - b[:, :1]) / 100.0   # us since block start
names = ["start", "prologue issued", "prologue landed", "k-loop done", "phase 0 out", "phase 1 out", "phase 2 out", "phase 3 out"]
print("%d blocks; median microseconds since block start:" % len(b))
for i, n in enumerate(names):
    print("  %-16s %8.2f   (p10 %.2f, p90 %.2f)" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
st = (b[:, 0] - t0) / 100.0
print("block start times (us since first): p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f; last block end %.1f"
      % tuple(list(np.percentile(st, [0, 25, 50, 75, 100])) + [(b[:, 7].max() - t0) / 100.0]))
