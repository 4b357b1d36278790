"""DD-PPO collectives: drop-in for the distributed part of audio_separation/rl/ppo/ddppo_utils.py (:117-190) and of
DecentralizedDistributedMixin (ppo.py:275-319).  One process per GPU, torch.distributed over RCCL ("nccl" on ROCm); the
same code runs over gloo on CPU for the world_size-2 tests.  The data path of the hot loop has exactly three collectives:

  broadcast_parameters   once, rank 0 -> all  (DDP construction, ppo.py:298-307)
  reduce_gradients       one flat sum all-reduce per backward (23.3 MB for the policy; replaces DDP's bucketed reducer,
                         ppo.py:313-319); the division by world size happens inside the optimizer step
  advantage statistics   two scalar all-reduces per update_pol (ddppo_utils.py:168-190)

Overlap with the backward (the reference's DDP reducer, ppo.py:286-319): the policy's gradient is cut into two buckets in
reverse-backward order -- recurrent encoder + heads, whose gradients are complete when the backward reaches the encoders'
features, and the three encoders.  ``GradReduceStep.early`` enqueues the first bucket's all-reduce on the side stream as soon
as its last weight gradient has been issued, under the encoders' backward; only the second bucket's all-reduce is exposed
before clip + Adam (bench.py: phases.grad_allreduce_exposed_ms).  Sums are element-wise, so the buckets give the flat buffer's values.

Overlap with the next phase: ``GradReduceStep`` runs the all-reduce + clip + Adam of the LAST mini-batch of an update on a side HIP
stream and hands back a fence; whoever next reads those parameters (Policy.act / get_value / evaluate_* for the policy,
Policy.get_monoFromMem* for the acoustic memory) makes its stream wait on the fence first.  The collective and the optimizer
step of update_pol's last epoch therefore run under the following update_sep passes / the bookkeeping of the next rollout,
and those of update_sep's last epoch under the separator passes that open the next rollout -- the update itself is unchanged
(gradients averaged before clip and step, every reader ordered after the step), which tests/test_distributed_cpu.py and
tests/test_gpu_trainer.py check by comparing weights with the synchronous schedule.

The local arithmetic (means, squared differences, normalisation) is injected as callables: HIP kernels in the product
(m2h.ops.advantages / adv_sqdiff / adv_apply), the CPU oracle in the gloo tests.
"""
import os

import torch
import torch.distributed as dist


def init_distrib(backend="nccl", device=None):
    """Env-var rendezvous of torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), the same
    variables init_distrib_slurm reads (ddppo_utils.py:131-147).  Returns (local_rank, world_rank, world_size)."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "8738")  # config/default.py:97
    if world_size > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=world_rank, world_size=world_size, **kw)
    return local_rank, world_rank, world_size


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_parameters(tensors, src=0):
    if world_size() == 1:
        return
    with torch.no_grad():
        for t in tensors:
            dist.broadcast(t.data, src)


_COLLECTIVE_LOG = None


def collective_log(enable=True):
    """Diagnostics (bench.py): while enabled every gradient all-reduce on a GPU tensor is bracketed by two timing events on the
    stream it is enqueued on (the compute stream, or GradReduceStep's side stream) and logged as (start, end, payload bytes, on the compute stream?).
    Returns the list being filled (None when disabled); the caller reads the events after a device synchronize."""
    global _COLLECTIVE_LOG
    _COLLECTIVE_LOG = [] if enable else None
    return _COLLECTIVE_LOG


def reduce_gradients(flat_grad, exposed=True):
    """Sum all-reduce of the flat gradient buffer (or a bucket of it); returns the scale (1/world) the optimizer applies before clipping.
    exposed: the collective is enqueued on the compute stream (it waits for it), not on the side stream (logged for bench.py)."""
    w = world_size()
    if w == 1:
        return 1.0
    log = _COLLECTIVE_LOG
    if log is not None and flat_grad.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_reduce(flat_grad)
        e1.record()
        log.append((e0, e1, flat_grad.numel() * flat_grad.element_size(), bool(exposed)))
    else:
        dist.all_reduce(flat_grad)
    return 1.0 / w


class GradReduceStep:
    """Gradient all-reduce + optimizer step of one backward, synchronous or deferred behind a fence.

    submit(flat_grad, step_fn, defer): ``step_fn(grad_scale)`` launches clip + Adam on the *current* stream.
      defer=False: all-reduce and step are enqueued on the caller's stream (the reference's order, ppo.py:313-319).
      defer=True : GPU tensors -- both are enqueued on a side stream that first waits for the caller's stream (the backward);
                   the caller's stream does not wait.  CPU tensors (gloo tests) -- the work is kept as a closure and executed
                   by fence(), i.e. as late as the schedule allows, so a reader that forgets its fence shows up as a mismatch.
    fence(): orders the current stream after the pending step (stream wait, no host sync); no-op when nothing is pending.
    """

    def __init__(self):
        self._side = None
        self._dev = None
        self._event = None
        self._lazy = None
        self._early_event = None
        self.deferred_steps = 0
        self.early_buckets = 0

    def _side_stream(self, dev):
        self._dev = dev
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def early(self, flat_bucket):
        """Sum all-reduce of a bucket whose gradients are complete while the rest of the backward is still to be enqueued: on the
        side stream, behind what the caller's stream holds so far; the caller's stream does not wait (submit() orders the
        optimizer step after it).  CPU tensors (gloo tests): reduced at once."""
        self.early_buckets += 1
        if not flat_bucket.is_cuda:
            reduce_gradients(flat_bucket)
            return
        dev = flat_bucket.device
        side = self._side_stream(dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        side.wait_event(ready)
        with torch.cuda.stream(side):
            reduce_gradients(flat_bucket, exposed=False)
            self._early_event = torch.cuda.Event()
            self._early_event.record(side)

    def pending(self):
        return self._event is not None or self._lazy is not None

    def submit(self, flat_grad, step_fn, defer=False):
        self.fence()  # at most one step in flight per parameter group; its buffers are about to be reused
        if not defer:
            scale = reduce_gradients(flat_grad)
            if self._early_event is not None:     # the early bucket's all-reduce (side stream) before the step reads the whole buffer
                ev, self._early_event = self._early_event, None
                torch.cuda.current_stream(self._dev).wait_event(ev)
            step_fn(scale)
            return
        self.deferred_steps += 1
        if not flat_grad.is_cuda:
            self._lazy = lambda: step_fn(reduce_gradients(flat_grad))
            return
        dev = flat_grad.device
        self._side_stream(dev)
        self._early_event = None   # (an early bucket ran on this same side stream: already ordered before what follows)
        backward_done = torch.cuda.Event()
        backward_done.record(torch.cuda.current_stream(dev))
        self._side.wait_event(backward_done)
        with torch.cuda.stream(self._side):
            step_fn(reduce_gradients(flat_grad, exposed=False))
            self._event = torch.cuda.Event()
            self._event.record(self._side)

    def fence(self):
        if self._lazy is not None:
            fn, self._lazy = self._lazy, None
            fn()
        if self._event is not None:
            ev, self._event = self._event, None
            torch.cuda.current_stream(self._dev).wait_event(ev)


class RolloutTracker:
    """The straggler pre-emption counter of the reference's DD-PPO loop (ppo_trainer.py:597-600, :769-782, :863): a key
    ``rollout_tracker/num_done`` in the job's key-value store.  Every rank adds 1 when its rollout is complete (:781-782); a rank still
    collecting stops early once it has made ``short_rollout_threshold`` of its steps AND more than ``sync_frac`` of the ranks are done
    (:775-780); world rank 0 puts the counter back to 0 after the update's statistics all-reduce (:862-863).  The shipped configs set
    short_rollout_threshold 1.0 (nearTarget.yaml:58): ``step >= num_steps`` never holds inside the loop, the store is then never read.
    store: the process group's own store (one process per GPU, torch.distributed) under the reference's prefix; without a process group a
    local counter (world size 1: nobody else to wait for)."""

    def __init__(self, world_size=1, world_rank=0, store=None):
        self.world_size, self.world_rank = int(world_size), int(world_rank)
        if store is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            store = dist.PrefixStore("rollout_tracker", _get_default_store())
        self._store = store
        self._local = 0
        self.reads = 0
        if self._store is not None:
            self._store.set("num_done", "0")           # :600 (every rank does, before the first rollout)

    def num_done(self):
        self.reads += 1
        return int(self._store.get("num_done")) if self._store is not None else self._local

    def rollout_done(self):
        if self._store is not None:
            self._store.add("num_done", 1)
        else:
            self._local += 1

    def reset(self):
        """World rank 0, after the update's all-reduce (:862-863)."""
        if self.world_rank == 0:
            if self._store is not None:
                self._store.set("num_done", "0")
            else:
                self._local = 0

    def should_preempt(self, step, num_steps, short_rollout_threshold, sync_frac):
        """:775-780, evaluated after step `step` (0-based) of the rollout: the store is read only once the threshold is reached."""
        if step < num_steps * short_rollout_threshold:
            return False
        return self.num_done() > sync_frac * self.world_size


def normalize_advantages_distributed(raw_adv, local_mean, sqdiff_fn, apply_fn, eps=1e-5):
    """_get_advantages_distributed (ppo.py:275-284): global mean = mean of per-rank means, global var = mean over ranks of the
    per-rank mean((A - mean)^2) (biased), A <- (A - mean) / (sqrt(var) + eps).
    local_mean: 1-element tensor; sqdiff_fn(adv, mean) -> 1-element tensor mean((adv-mean)^2); apply_fn(adv, mean, var, eps)."""
    w = world_size()
    mean = local_mean.clone()
    if w > 1:
        dist.all_reduce(mean)
        mean /= w
    var = sqdiff_fn(raw_adv, mean)
    if w > 1:
        dist.all_reduce(var)
        var /= w
    return apply_fn(raw_adv, mean, var, eps)


def all_reduce_stats(t):
    """The logging all-reduces of ppo_trainer.py:839-860 fused into one call."""
    if world_size() > 1:
        dist.all_reduce(t)
    return t
