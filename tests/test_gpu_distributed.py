"""GPU, one rank: the DD-PPO cycle with a live RCCL process group -- the flat-gradient all-reduce issued on the side stream
(deferred last step of every update) while the compute stream replays the rollout / update HIP graphs, captured with RCCL's
watchdog thread running.  One GPU cannot host two ranks, so the group has world size 1 and the all-reduce is forced on; the
2-rank protocol itself is covered on CPU (tests/test_distributed_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "move2hear-active-av-separation_amd"))
import numpy as np, torch, torch.distributed as dist
from m2h import synthetic
from m2h.rl.ppo import ddppo_utils as D
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
calls = [0]
def reduce_gradients(flat):   # world size 1 would skip the collective: issue it anyway (sum over one rank = identity)
    calls[0] += 1
    dist.all_reduce(flat)
    return 1.0
D.reduce_gradients = reduce_gradients
out = []
for graphs, overlap in ((False, False), (True, True)):
    cfg = near_target_config(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=4,
                             use_hip_graphs=graphs, overlap_grad_reduce=overlap)
    tr = PPOTrainer(cfg, dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 5).items()})
    for c in range(3):
        torch.manual_seed(40 + c)
        res = tr.train_cycle()
    stats = tr.all_reduce_stats()
    out.append(({k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}, res["pol_losses"], res["sep_losses"]))
    if graphs:
        assert tr._graph_state is not None and len(tr._graph_state.graphs) >= 2 and tr.agent._pol_graph.graph is not None
        assert tr.agent._reducers["pol"].deferred_steps == 6
(wa, pa, sa), (wb, pb, sb) = out
assert pa == pb and sa == sb, (pa, pb, sa, sb)
for k in wa:
    assert torch.equal(wa[k], wb[k]), k
assert calls[0] == 2 * 3 * 2 * 2 * 2   # 2 runs x 3 cycles x (2 update_pol + 2 update_sep) x 2 epochs
dist.destroy_process_group()
print("OK", calls[0])
'''


def test_cycle_with_rccl_group_side_stream_allreduce_and_graphs():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "port": port}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
