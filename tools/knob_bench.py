#!/usr/bin/env python3
"""bench.py under one m2h_tuning_set knob (tuning tool): python tools/knob_bench.py KNOB VALUE [bench.py arguments ...]
e.g. tools/knob_bench.py 24 4096 --ddppo-cycles 2 --no-far-target --train-steps 0   (measured: the skinny gather kernel's pixel limit
1024 / 4096 / 16384 leaves the DD-PPO cycle at 9.5-9.7 K env-steps/s)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "move2hear-active-av-separation_amd"))
knob, val = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0]] + sys.argv[3:]
from m2h import functional as MF, ops
MF.carry_tuning(True)   # the knobs are thread-local: backward passes run on autograd's thread
ops.debug_set(knob, val)
import bench
bench.main()
