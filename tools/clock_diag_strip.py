#!/usr/bin/env python3
"""Diagnostic (not product): builds a copy of libm2h with -DM2H_CLOCK_DIAG and prints where the waves of the strip-walker kernels
(csrc/conv_strip.hip) spend their shader-clock cycles, per step on average, at the benchmark shape.
usage: [M2H_STRIP_VARIANTS=conv1,conv1m,last32,last16] python tools/clock_diag_strip.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402

from m2h import _lib  # noqa: E402

diag = "/tmp/libm2h_diag.so"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DM2H_CLOCK_DIAG", "-DM2H_STRIP_DBG=%d" % int(os.environ.get("M2H_STRIP_DBG", "0")),
       "-I" + _lib.INCLUDE, "-I" + _lib.CSRC]
cmd += [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES] + ["-o", diag]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = diag
import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
import strip_bench  # noqa: E402

lib = _lib.load()
lib.m2h_diag_read_clocks_strip.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["load issue", "MFMA loop", "epilogue", "barrier 1", "head / copy-out", "ring store", "barrier 2", "copy-out (last)", "prologue (per job)", "whole kernel"]
for variant in (os.environ.get("M2H_STRIP_VARIANTS", "conv1,conv1m,last32,last16")).split(","):
    sys.argv = [sys.argv[0], "--reps", "3", "--only", variant]
    strip_bench.main()
    torch.cuda.synchronize()
    buf = np.zeros((1024, 8, 10), np.uint64)
    lib.m2h_diag_read_clocks_strip(buf.ctypes.data, 1024)
    nw = 8 if variant == "last32" else 4
    nblk = 256 if variant == "last32" else 512
    b = buf[:nblk, :nw].astype(np.float64)
    jobs = 256 * 4 / nblk
    steps = jobs * 16
    print("%s: cycles per step, median over waves (p10 .. p90); %d workgroups x %d waves, %g jobs x 16 steps each" % (variant, nblk, nw, jobs))
    for i, n in enumerate(names):
        div = steps if i < 8 else (jobs if i == 8 else 1.0)
        v = b[:, :, i].reshape(-1) / div
        print("  %-20s %10.0f   (%.0f .. %.0f)" % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
