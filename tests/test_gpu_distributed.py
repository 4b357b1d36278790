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
def reduce_gradients(flat, exposed=True):   # world size 1 would skip the collective: issue it anyway (sum over one rank = identity)
    calls[0] += 1
    dist.all_reduce(flat)
    return 1.0
D.reduce_gradients = reduce_gradients
out = []
for graphs, overlap in ((False, False), (True, True)):
    cfg = near_target_config(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=4,
                             use_hip_graphs=graphs, overlap_grad_reduce=overlap, bucketed_grad_reduce=overlap)
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
        assert tr.agent._reducers["pol"].deferred_steps == 6 and tr.agent._reducers["pol"].early_buckets == 3 * 2 * 2
(wa, pa, sa), (wb, pb, sb) = out
assert pa == pb and sa == sb, (pa, pb, sa, sb)
for k in wa:
    assert torch.equal(wa[k], wb[k]), k
assert calls[0] == 3 * 2 * 2 * 2 + 3 * (2 * 2 + 2) * 2   # 3 cycles x (2 update_pol + 2 update_sep) x 2 epochs; the second run: two buckets per policy epoch
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


BUCKETS = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "move2hear-active-av-separation_amd"))
import numpy as np, torch, torch.distributed as dist
from m2h import graphs, synthetic
from m2h.rl.ppo import ddppo_utils as D
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
log = []          # host enqueue order of the cycle: ("replay", graph) and ("reduce", bytes, exposed, on the compute stream?, start event)
main = torch.cuda.current_stream(dev)
def reduce_gradients(flat, exposed=True):     # world size 1 would skip the collective: issue it anyway, live, on whatever stream is current
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    dist.all_reduce(flat)
    log.append(("reduce", flat.numel() * 4, bool(exposed), torch.cuda.current_stream(dev) == main, e))
    return 1.0
D.reduce_gradients = reduce_gradients
real_replay = graphs.replay
def replay(g):
    real_replay(g)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    log.append(("replay", g, e))
graphs.replay = replay
import m2h.rl.ppo.ppo as P, m2h.rl.ppo.ppo_trainer as T
assert P.graphs is graphs and T.graphs is graphs
# the REFERENCE schedule's counts (nearTarget.yaml: 6 updates per cycle, 4 epochs, 1 mini-batch) on a small batch, the N > 1 gradient schedule forced on
cfg = near_target_config(NUM_PROCESSES=3, num_steps=4, MAX_EPISODE_STEPS=4, use_hip_graphs=True, overlap_grad_reduce=True, bucketed_grad_reduce=True)
assert cfg.num_updates_per_cycle == 6 and cfg.ppo_epoch == 4 and cfg.num_mini_batch == 1
tr = PPOTrainer(cfg, dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 5).items()})
tr.train_cycle()          # warm-up: the first policy update runs kernel by kernel, every graph is captured
torch.cuda.synchronize()
del log[:]
tr.train_cycle()
torch.cuda.synchronize()
ag = tr.agent
pg = ag._pol_graph
assert pg is not None and pg.bucketed and pg.graph is not None and pg.graph_rest is not None and ag._sep_graph.graph is not None
opt = ag.optimizer_pol
tail_bytes = ag.optimizer_pol.grad_bucket(ag._tail_bucket_params()).numel() * 4
pol_bytes, mem_bytes = opt.grad_buffer().numel() * 4, ag.optimizer_sep.grad_buffer().numel() * 4
enc_bytes = pol_bytes - tail_bytes
assert len({tail_bytes, enc_bytes, mem_bytes}) == 3 and tail_bytes > 0 and enc_bytes > 0
red = [x for x in log if x[0] == "reduce"]
count = lambda nbytes, exposed: sum(1 for x in red if x[1] == nbytes and x[2] == exposed)
# DESIGN section 6: per cycle 24 tail buckets (recurrent encoder + heads) hidden under the encoders' backward; of the 24 encoder buckets and the
# 24 memory gradients the 3 first epochs' of each update are exposed (18 + 18), the last epoch's go with the deferred step (6 + 6 = the 12 last steps)
assert count(tail_bytes, False) == 24 and count(tail_bytes, True) == 0
assert count(enc_bytes, True) == 18 and count(enc_bytes, False) == 6
assert count(mem_bytes, True) == 18 and count(mem_bytes, False) == 6
assert len(red) == 72 and sum(1 for x in red if x[2]) == 36
# exposed collectives sit on the compute stream, hidden ones on a side stream
assert all(x[3] == x[2] for x in red), [(x[1], x[2], x[3]) for x in red if x[3] != x[2]][:4]
# every policy epoch: graph (forward + losses + backward to the encoders' features) -> the tail bucket's collective ENQUEUED -> the encoders' backward
# graph -> the encoder bucket.  Host order = stream order; on the device the tail collective has started before the encoders' backward ended.
seq = [x for x in log if (x[0] == "replay" and x[1] in (pg.graph, pg.graph_rest)) or (x[0] == "reduce" and x[1] in (tail_bytes, enc_bytes))]
assert len(seq) == 24 * 4
for i in range(0, len(seq), 4):
    a, b, c, d = seq[i:i + 4]
    assert a[0] == "replay" and a[1] is pg.graph and b[0] == "reduce" and b[1] == tail_bytes and not b[2], (i, a[:2], b[:3])
    assert c[0] == "replay" and c[1] is pg.graph_rest and d[0] == "reduce" and d[1] == enc_bytes, (i, c[:2], d[:3])
    assert b[4].elapsed_time(c[2]) >= 0.0          # tail collective's start event precedes the end of the encoders' backward graph
    assert a[2].elapsed_time(b[4]) >= 0.0          # ... and follows the first graph (its gradients are complete)
assert ag._reducers["pol"].early_buckets >= 24 and ag._reducers["pol"].deferred_steps >= 6 and ag._reducers["mem"].deferred_steps >= 6
dist.destroy_process_group()
print("OK", len(red))
'''


def test_bucketed_schedule_under_graphs_issues_the_collectives_design_section_6_predicts():
    """One rank, live RCCL, HIP graphs, the N > 1 gradient schedule forced on, the reference's update counts: per cycle 24 tail-bucket
    all-reduces on the side stream, each enqueued between the epoch's two graphs (before the encoders' backward), 18 + 18 exposed
    collectives on the compute stream, 6 + 6 riding with the deferred last steps -- so that the first multi-GPU run cannot silently fall
    back to the flat schedule (rl/ppo/ppo.py:286-319 is what this schedule replaces)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", BUCKETS % {"root": ROOT, "port": port}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK 72" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


TWO_RANK = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "move2hear-active-av-separation_amd"))
import numpy as np, torch, torch.distributed as dist
from m2h import synthetic
from m2h.rl.ppo import ddppo_utils as D
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
dev = torch.device("cuda", 0)            # both ranks share the box's one GPU; the collectives run over gloo
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=2)
cfg = near_target_config(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=4)
tr = PPOTrainer(cfg, dev, world_rank=rank, world_size=2)
tr.setup()
assert tr.agent._world == 2 and tr.agent._overlap()
# different initial weights per rank: init_distributed must have replaced rank 1's by rank 0's
w0 = {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}
for c in range(3):
    torch.manual_seed(100 * rank + c)     # different action samples per rank: only the averaged gradients keep the replicas equal
    res = tr.train_cycle()
stats = tr.all_reduce_stats()
sd = {k: v.detach().cpu() for k, v in tr.actor_critic.state_dict().items()}
flat = torch.cat([v.reshape(-1).double() for k, v in sorted(sd.items()) if v.dtype == torch.float32])
init = torch.cat([v.reshape(-1).double() for k, v in sorted(w0.items()) if v.dtype == torch.float32])
mine = torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(flat.numel(), dtype=torch.float64) %% 7).sum(), init.sum()])
both = [torch.zeros_like(mine) for _ in range(2)]
dist.all_gather(both, mine)
assert torch.equal(both[0], both[1]), (both[0], both[1])                     # identical replicas after three cycles, identical start
assert float((flat - init).abs().sum()) > 0                                  # and they did train
obs_seed = tr.rollouts_pol.observations["rgb"].double().sum().cpu()
seeds = [torch.zeros_like(obs_seed) for _ in range(2)]
dist.all_gather(seeds, obs_seed)
assert not torch.equal(seeds[0], seeds[1])                                   # each rank rolled out its own environments (seed + 3*rank)
assert tr.agent._reducers["pol"].deferred_steps == 6 and tr._graph_state is not None and tr.agent._pol_graph.graph is not None
assert tr.agent._bucketed() and tr.agent._pol_graph.graph_rest is not None and tr.agent._reducers["pol"].early_buckets == 3 * 2 * 2
assert float(stats[1]) == 2 * 3 * 3 * 2                                      # finished episodes summed over both ranks (3 cycles x 2 x 3 envs x 2 ranks)
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_two_ranks_train_identical_replicas_with_graphs_and_overlap():
    """Two DD-PPO ranks (sharing the one GPU, collectives over gloo): rank-0 parameter broadcast, flat-gradient all-reduce with
    the deferred side-stream step, distributed advantage statistics, HIP-graph rollouts / update epochs -- the replicas must
    stay bit-identical while each rank samples its own actions in its own environments."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", TWO_RANK % {"root": ROOT, "port": port}, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK_OK %d" % r) in o, o[-4000:]
