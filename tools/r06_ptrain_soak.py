"""Soak of the passive training step (batch 64, HIP-graph replays with the deferred weight gradients): 400 steps, loss trend, memory."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "move2hear-active-av-separation_amd"))
import torch
from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config

dev = torch.device("cuda", 0)
tr = PassiveTrainer(passive_config(BATCH_SIZE=64), dev)
tr.setup()
tr.actor_critic.train()
batches = [tr.feeders["train"].batch() for _ in range(8)]
for block in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = torch.zeros(2, device=dev)
    for i in range(100):
        b, m = tr.train_batch(*batches[i % 8])
        acc += torch.stack((b, m))
    torch.cuda.synchronize()
    print("steps %d-%d: %.3f ms/step, reserved %.2f GB, allocated %.2f GB, mean losses (bin, mono) %s" % (
        100 * block, 100 * block + 99, (time.perf_counter() - t0) * 10, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30,
        [round(x, 5) for x in (acc / 100).tolist()]))
