"""CPU, 2 processes over gloo: the collective protocol of the DD-PPO path (m2h/rl/ppo/ddppo_utils.py) -- parameter
broadcast, flat gradient all-reduce + averaging, distributed advantage normalisation -- with the local arithmetic supplied
by the CPU oracle (the HIP kernels that supply it in the product are covered by the -m gpu tests)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "move2hear-active-av-separation_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import m2h_oracle as O
    from m2h.rl.ppo import ddppo_utils as D
    lr, wr, ws = D.init_distrib("gloo")
    assert (lr, wr, ws) == (rank, rank, world) and D.world_size() == world
    # 1. broadcast: every rank ends with rank 0's parameters
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g)]
    D.broadcast_parameters(params)
    ref = [torch.randn(7, 5, generator=torch.Generator().manual_seed(100)), None]
    ok_b = torch.equal(params[0], ref[0])
    # 2. flat gradient all-reduce + 1/world scaling == mean of the per-rank gradients
    grads = [torch.full((1000,), float(r + 1)) + torch.arange(1000.0) * (r + 1) for r in range(world)]
    flat = grads[rank].clone()
    scale = D.reduce_gradients(flat)
    ok_g = torch.allclose(flat * scale, sum(grads) / world)
    # 3. distributed advantage normalisation == the reference formula evaluated on all ranks' data
    advs = [torch.randn(20, 14, 1, generator=torch.Generator().manual_seed(7 + r)) * (1 + r) + r for r in range(world)]
    mine = advs[rank].clone()
    out = D.normalize_advantages_distributed(
        mine, mine.mean().reshape(1), lambda a, m: (a - m).pow(2).mean().reshape(1), lambda a, m, v, e: (a - m) / (v.sqrt() + e), 1e-5)
    expect = O.get_advantages_distributed(advs)[rank]
    ok_a = torch.allclose(out, expect, atol=1e-6)
    # 4. fused stats all-reduce
    t = D.all_reduce_stats(torch.tensor([1.0 + rank, 2.0]))
    ok_s = torch.allclose(t, torch.tensor([sum(1.0 + r for r in range(world)), 2.0 * world]))
    # 5. deferred all-reduce + step (GradReduceStep): a toy DD-PPO schedule -- "rollouts" read the parameters through the fence,
    #    each update = 2 epochs whose last step is deferred -- ends with the weights of the synchronous schedule, and the
    #    deferral is real (before the fence the parameters are still the old ones)
    def run(defer_last):
        p = torch.linspace(-1, 1, 64).clone()
        red = D.GradReduceStep()
        seen, late = [], True
        for upd in range(3):
            red.fence()                                   # reader: the rollout
            seen.append(p.clone())
            for ep in range(2):
                red.fence()                               # reader: evaluate_actions
                grad = torch.sin(p * (upd + 1)) * (rank + 1) + ep

                def step(scale, grad=grad):
                    p.sub_(0.1 * grad * scale)
                before = p.clone()
                red.submit(grad, step, defer=defer_last and ep == 1)
                if defer_last and ep == 1:
                    late = late and torch.equal(p, before) and red.pending()
        red.fence()
        return p, seen, late, red.deferred_steps
    p_sync, seen_sync, _, n_sync = run(False)
    p_def, seen_def, late, n_def = run(True)
    ok_d = torch.equal(p_sync, p_def) and all(torch.equal(a, b) for a, b in zip(seen_sync, seen_def)) and late and (n_sync, n_def) == (0, 3)
    # 6. bucketed reduction (GradReduceStep.early): the tail bucket of the flat gradient is all-reduced first, the head bucket by
    #    submit() -- element-wise sums, so the step sees the values of the one flat all-reduce, synchronous or deferred
    def run_buckets(bucketed, defer):
        p = torch.linspace(-1, 1, 64).clone()
        red = D.GradReduceStep()
        for upd in range(3):
            red.fence()
            grad = torch.cos(p * (upd + 2)) * (rank + 1)

            def step(scale, grad=grad):
                p.sub_(0.05 * grad * scale)
            if bucketed:
                red.early(grad[40:])          # (the step function reads the whole buffer: both buckets must be summed by then)
                red.submit(grad[:40], step, defer=defer)
            else:
                red.submit(grad, step, defer=defer)
        red.fence()
        return p, red.early_buckets
    p_flat, n0 = run_buckets(False, False)
    p_b, n1 = run_buckets(True, False)
    p_bd, n2 = run_buckets(True, True)
    ok_e = torch.equal(p_flat, p_b) and torch.equal(p_flat, p_bd) and (n0, n1, n2) == (0, 3, 3)
    q.put((rank, ok_b, ok_g, ok_a, ok_s, ok_d, ok_e))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_protocol():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert all(r[1:]), r


def _tracker_rank(rank, port, q):
    import torch.distributed as dist
    from m2h.rl.ppo import ddppo_utils as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    tr = D.RolloutTracker(2, rank)
    dist.barrier()
    out = {}
    # below the threshold the store is not even read
    out["early"] = (tr.should_preempt(3, 20, 0.5, 0.4), tr.reads)
    if rank == 1:
        tr.rollout_done()                     # rank 1 finishes its rollout first
    dist.barrier()
    # rank 0, still collecting at step 10 of 20 (threshold 0.5): 1 rank done > 0.4 * 2 -> stop early; with sync_frac 0.6 (1 > 1.2 is false) go on
    out["preempt"] = (tr.should_preempt(10, 20, 0.5, 0.4), tr.should_preempt(10, 20, 0.5, 0.6), tr.num_done())
    dist.barrier()
    if rank == 0:
        tr.rollout_done()
    dist.barrier()
    out["both"] = tr.num_done()
    dist.barrier()
    tr.reset()                                # only world rank 0 writes
    dist.barrier()
    out["after_reset"] = tr.num_done()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_rollout_tracker_preempts_stragglers_as_the_reference_counter_does():
    """ppo_trainer.py:597-600, :769-782, :862-863 over the process group's own store, two gloo ranks: every rank adds when its rollout is
    done, a rank still collecting stops once it is past short_rollout_threshold and more than sync_frac of the ranks wait, rank 0 resets."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tracker_rank, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["early"] == (False, 0)
        assert res[r]["preempt"] == (True, False, 1)
        assert res[r]["both"] == 2 and res[r]["after_reset"] == 0
    # without a process group: a local counter, world size 1 -- nobody to wait for, the own add comes after the loop
    from m2h.rl.ppo import ddppo_utils as D
    t = D.RolloutTracker(1, 0)
    assert not t.should_preempt(19, 20, 0.5, 0.6)
    t.rollout_done()
    assert t.num_done() == 1 and t.should_preempt(10, 20, 0.5, 0.6)
    t.reset()
    assert t.num_done() == 0
