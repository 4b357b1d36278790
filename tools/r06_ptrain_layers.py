"""Per-launch table of one passive training step (batch 64), kernel by kernel with HIP events around every libm2h conv-engine call:
name, label of the kernel the dispatch took, M x N x K, microseconds, GB/s on the launch's own bytes (weights + activations)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "move2hear-active-av-separation_amd"))
import torch
from m2h import ops
from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config

dev = torch.device("cuda", 0)
from m2h import functional as MF
MF.carry_tuning(True)
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("=")
    ops.debug_set(int(k), int(v))
tr = PassiveTrainer(passive_config(BATCH_SIZE=64, use_hip_graphs=False), dev)
tr.setup()
tr.actor_critic.train()
batch = tr.feeders["train"].batch()
for _ in range(3):
    tr.train_batch(*batch)
torch.cuda.synchronize()
sink = []
ops.set_timing(sink)
tr.train_batch(*batch)
torch.cuda.synchronize()
ops.set_timing(None)
tot = 0.0
for name, meta, e0, e1 in sink:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    m = meta or {}
    by = m.get("bytes")
    print("%-22s %-58s M %6s N %5s K %6s  %7.1f us  %s" % (name, (m.get("label") or "")[:58], m.get("M", ""), m.get("N", ""), m.get("K", ""), us,
                                                        ("%6.0f GB/s" % (by / us / 1e3)) if by else ""))
print("timed launches: %d, sum %.1f us" % (len(sink), tot))
