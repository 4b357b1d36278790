"""Rollout storage on the GPU: drop-in for audio_separation/common/rollout_storage.py.

Same class names, constructor arguments, attribute names and methods (insert, after_update, compute_returns,
recurrent_generator, to).  Differences in mechanism, not in results:
  * compute_returns runs one HIP scan kernel instead of a python loop of ~6 tiny launches per step (:155-180);
  * recurrent_generator draws ``torch.randperm(num_envs)`` on the CPU generator exactly as the reference does (:197, :406)
    (bit-exact env order for a given seed) and gathers each storage tensor with one HBM-bound kernel instead of
    per-env python slicing + stack + flatten.
"""
from collections import defaultdict

import torch

from .. import ops


def _roll_rows(tensors):
    """row[0] <- row[-1] for every storage tensor: ONE batched device copy (m2h_rows_copy) instead of one launch per tensor (13 per
    after_update); on the CPU (tests of the host logic) the plain copies."""
    tensors = [t for t in tensors if t is not None]
    if tensors and all(t.is_cuda for t in tensors):
        from .. import ops
        idx = _roll_rows.idx.get(tensors[0].device)
        if idx is None:
            idx = _roll_rows.idx[tensors[0].device] = torch.zeros(1, dtype=torch.int64, device=tensors[0].device)
        ops.rows_copy([(t[-1], t[0], -1, -1) for t in tensors], idx)
    else:
        for t in tensors:
            t[0].copy_(t[-1])


_roll_rows.idx = {}


class RolloutStoragePol:
    def __init__(self, num_steps, num_envs, observation_space, recurrent_hidden_state_size, num_recurrent_layers=1):
        self.observations = {}
        for sensor in observation_space.spaces:
            self.observations[sensor] = torch.zeros(num_steps + 1, num_envs, *observation_space.spaces[sensor].shape)
        self.recurrent_hidden_states_pol = torch.zeros(num_steps + 1, num_recurrent_layers, num_envs, recurrent_hidden_state_size)
        assert "gt_mono_comps" in observation_space.spaces
        f, t = observation_space.spaces["gt_mono_comps"].shape[:2]
        self.pred_binSepMasks = torch.zeros(num_steps, num_envs, f, t, 2)
        self.pred_mono = torch.zeros(num_steps, num_envs, f, t, 1)
        self.prev_pred_monoFromMem = torch.zeros(num_steps + 1, num_envs, f, t, 1)
        self.rewards = torch.zeros(num_steps, num_envs, 1)
        self.value_preds = torch.zeros(num_steps + 1, num_envs, 1)
        self.returns = torch.zeros(num_steps + 1, num_envs, 1)
        self.action_log_probs = torch.zeros(num_steps, num_envs, 1)
        self.actions = torch.zeros(num_steps, num_envs, 1).long()
        self.masks = torch.ones(num_steps + 1, num_envs, 1)
        self.num_steps = num_steps
        self.step = 0

    _TENSORS = ("recurrent_hidden_states_pol", "pred_binSepMasks", "pred_mono", "prev_pred_monoFromMem", "rewards", "value_preds",
                "returns", "action_log_probs", "actions", "masks")

    def to(self, device):
        for sensor in self.observations:
            self.observations[sensor] = self.observations[sensor].to(device)
        for name in self._TENSORS:
            setattr(self, name, getattr(self, name).to(device))

    def _rows(self, observations, recurrent_hidden_states_pol, actions, action_log_probs, values, rewards, masks,
              pred_binSepMasks=None, pred_mono=None, pred_monoFromMem=None):
        """(storage tensor, row offset relative to step, value) of one insert (reference :68-96)."""
        rows = [(self.observations[sensor], 1, observations[sensor]) for sensor in observations]
        rows += [(self.recurrent_hidden_states_pol, 1, recurrent_hidden_states_pol), (self.pred_binSepMasks, 0, pred_binSepMasks),
                 (self.pred_mono, 0, pred_mono), (self.prev_pred_monoFromMem, 1, pred_monoFromMem), (self.rewards, 0, rewards),
                 (self.value_preds, 0, values), (self.actions, 0, actions), (self.action_log_probs, 0, action_log_probs),
                 (self.masks, 1, masks)]
        return rows

    def insert(self, *args, **kwargs):
        for dst, off, v in self._rows(*args, **kwargs):
            dst[self.step + off].copy_(v)
        self.advance()

    def insert_items(self, slots, *args, **kwargs):
        """The same insert as a list of m2h.ops.rows_copy items addressing the rows through device-resident indices
        (slots = positions of `step` and `step + 1` in the index tensor): for the rollout step replayed from a HIP graph,
        where the host counter cannot be baked into addresses.  The host counter is left to the caller (``advance``)."""
        return [(v.contiguous().view(dst.shape[1:]), dst, -1, slots[off]) for dst, off, v in self._rows(*args, **kwargs)]

    def advance(self):
        self.step = (self.step + 1) % self.num_steps

    def after_update(self):
        _roll_rows(list(self.observations.values()) + [self.recurrent_hidden_states_pol, self.prev_pred_monoFromMem, self.masks])

    def compute_returns(self, next_value, use_gae, gamma, tau):
        ops.gae_returns(self.rewards, self.value_preds, self.masks, next_value.contiguous(), self.returns, use_gae, gamma, tau)

    def recurrent_generator(self, advantages, num_mini_batch):
        num_processes = self.rewards.size(1)
        assert num_processes >= num_mini_batch, (
            "Trainer requires the number of processes ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(num_processes, num_mini_batch))
        num_envs_per_batch = num_processes // num_mini_batch
        perm = torch.randperm(num_processes)  # CPU generator, as the reference
        dev = self.rewards.device
        # One mini-batch = every environment, merely re-ordered: the update is a mean over the batch and the GRU sequences are
        # per environment, so the order only changes fp summation order.  With full_batch_views the physical gather (hundreds
        # of MB per epoch) is skipped and the batch is a view of the storage; the permutation is still drawn (RNG state).
        ident = bool(getattr(self, "full_batch_views", False)) and num_mini_batch == 1
        for start_ind in range(0, num_processes, num_envs_per_batch):
            idx = None if ident else perm[start_ind:start_ind + num_envs_per_batch].to(dev)
            take = lambda t: ops.take_envs(t, idx, ident)  # noqa: E731
            observations_batch = defaultdict(list)
            for sensor in self.observations:
                observations_batch[sensor] = take(self.observations[sensor][:-1])
            # hidden state: [layers, N_sel, H] from step 0
            hs = self.recurrent_hidden_states_pol[0]  # [layers, N, H] at step 0
            recurrent_hidden_states_pol_batch = hs if ident else ops.gather_envs(hs, idx).view(hs.size(0), idx.numel(), hs.size(2))
            yield (
                observations_batch,
                recurrent_hidden_states_pol_batch,
                take(self.pred_binSepMasks),
                take(self.pred_mono),
                take(self.prev_pred_monoFromMem[1:]),
                take(self.value_preds[:-1]),
                take(self.returns[:-1]),
                take(advantages.contiguous()),
                take(self.actions),
                take(self.action_log_probs),
                take(self.masks[:-1]),
            )


class RolloutStorageSep:
    def __init__(self, num_steps, num_envs, observation_space):
        self.observations = {}
        for sensor in observation_space.spaces:
            self.observations[sensor] = torch.zeros(num_steps + 1, num_envs, *observation_space.spaces[sensor].shape)
        assert "gt_mono_comps" in observation_space.spaces
        f, t = observation_space.spaces["gt_mono_comps"].shape[:2]
        self.prev_pred_monoFromMem = torch.zeros(num_steps + 1, num_envs, f, t, 1)
        self.masks = torch.ones(num_steps + 1, num_envs, 1)
        self.num_steps = num_steps
        self.step = 0
        self.generation = 0  # bumped whenever stored observations change (keys PPO's separator-output cache)
        self.row0_only_since = None  # generation before the last bump that changed nothing but row 0 (after_update)
        # Optional: the frozen, eval-mode separators' outputs for every stored observation, written by the trainer's rollout step
        # (which computes them anyway, for the reward and for the next step) so that update_sep need not run the U-Nets over the
        # buffer again (the reference re-runs them 24 x per cycle under no_grad: ppo.py:184-195).  Rows are valid once
        # written since enable / invalidate; PPO.update_sep falls back to computing them when any needed row is not.
        self.pred_binSepMasks = None
        self.pred_mono = None
        self._pred_rows_valid = None

    def to(self, device):
        for sensor in self.observations:
            self.observations[sensor] = self.observations[sensor].to(device)
        self.prev_pred_monoFromMem = self.prev_pred_monoFromMem.to(device)
        self.masks = self.masks.to(device)
        if self.pred_mono is not None:
            self.pred_binSepMasks, self.pred_mono = self.pred_binSepMasks.to(device), self.pred_mono.to(device)
        self.generation += 1

    def touch(self):
        """Call after ANY write to the stored tensors that does not go through insert / advance / after_update / to (a direct
        ``observations[s][row].copy_(...)``, the reference's reset pattern ppo_trainer.py:669-671): PPO's separator-output cache, the sliced
        memory input and the after_update short-cut key on ``generation`` and would otherwise keep serving the old contents."""
        self.generation += 1
        self.row0_only_since = None

    def enable_separator_outputs(self):
        """Allocate the per-row separator outputs (see __init__); every row starts invalid."""
        ref = self.prev_pred_monoFromMem
        self.pred_binSepMasks = torch.zeros(tuple(ref.shape[:-1]) + (2,), device=ref.device)
        self.pred_mono = torch.zeros_like(ref)
        self._pred_rows_valid = [False] * (self.num_steps + 1)

    def invalidate_separator_outputs(self):
        """The separators' weights changed (checkpoint load): stored outputs no longer belong to them."""
        if self._pred_rows_valid is not None:
            self._pred_rows_valid = [False] * (self.num_steps + 1)

    def store_separator_outputs(self, row, pred_binSepMasks, pred_mono):
        """Outputs of the observation stored in `row`, computed outside an insert (the first step after a reset of the chain)."""
        if self.pred_mono is not None:
            self.pred_binSepMasks[row].copy_(pred_binSepMasks)
            self.pred_mono[row].copy_(pred_mono)
            self._pred_rows_valid[row] = True
            self.touch()

    def stored_separator_outputs(self):
        """(pred_binSepMasks, pred_mono) of rows 0 .. T-1 -- what update_sep's batch is made of -- or None when any is missing."""
        if self._pred_rows_valid is None or not all(self._pred_rows_valid[:-1]):
            return None
        return self.pred_binSepMasks[:-1], self.pred_mono[:-1]

    def _rows(self, observations, masks, pred_monoFromMem=None, pred_binSepMasks=None, pred_mono=None):
        rows = [(self.observations[sensor], observations[sensor]) for sensor in observations] + \
            [(self.prev_pred_monoFromMem, pred_monoFromMem), (self.masks, masks)]
        if self.pred_mono is not None and pred_mono is not None:   # separator outputs OF THE INSERTED observation
            rows += [(self.pred_binSepMasks, pred_binSepMasks), (self.pred_mono, pred_mono)]
        return rows

    def insert(self, observations, masks, pred_monoFromMem=None, pred_binSepMasks=None, pred_mono=None):
        for dst, v in self._rows(observations, masks, pred_monoFromMem, pred_binSepMasks, pred_mono):
            dst[self.step + 1].copy_(v)
        self.advance(with_preds=pred_mono is not None)

    def insert_items(self, slot, observations, masks, pred_monoFromMem=None, pred_binSepMasks=None, pred_mono=None):
        """See RolloutStoragePol.insert_items; slot = position of `step + 1` in the index tensor."""
        return [(v.contiguous().view(dst.shape[1:]), dst, -1, slot)
                for dst, v in self._rows(observations, masks, pred_monoFromMem, pred_binSepMasks, pred_mono)]

    def advance(self, with_preds=False):
        """with_preds: the insert this advance belongs to carried the inserted observation's separator outputs."""
        if self._pred_rows_valid is not None:
            self._pred_rows_valid[self.step + 1] = bool(with_preds)
        self.generation += 1
        self.step = (self.step + 1) % self.num_steps

    def after_update(self):
        # NB: copies obs[-1] -> obs[0]; when the buffer has just been refilled (step wrapped to 0) obs[0] already holds the
        # same tensor the policy storage carried over, so re-running after_update between the 6 sub-updates of a cycle does
        # not change what is stored: compare before bumping the generation would need a sync, so track it by step instead.
        changed = self.step != getattr(self, "_last_after_update_step", None) or self.generation != getattr(self, "_last_after_update_gen", None)
        if not changed:
            return   # nothing was inserted since the last call: row 0 already holds the last row (the cycle's separator updates 2-6: 13 copies each)
        _roll_rows(list(self.observations.values()) + [self.prev_pred_monoFromMem, self.masks] +
                   ([self.pred_binSepMasks, self.pred_mono] if self.pred_mono is not None else []))
        if self.pred_mono is not None:
            self._pred_rows_valid[0] = self._pred_rows_valid[-1]
        if changed:
            # only row 0 of the stored observations changed: a cache built for generation g - 1 needs row 0 refreshed, not rebuilt
            self.row0_only_since = self.generation
            self.generation += 1
            self._last_after_update_gen = self.generation
            self._last_after_update_step = self.step

    def recurrent_generator(self, num_mini_batch, with_perm=False, sensors=None):
        """sensors: optional subset of observation names to materialise (the separator update needs three of them)."""
        num_processes = self.masks.size(1)
        assert num_processes >= num_mini_batch
        num_envs_per_batch = num_processes // num_mini_batch
        perm = torch.randperm(num_processes)
        dev = self.masks.device
        ident = bool(getattr(self, "full_batch_views", False)) and num_mini_batch == 1  # see RolloutStoragePol.recurrent_generator
        for start_ind in range(0, num_processes, num_envs_per_batch):
            idx = None if ident else perm[start_ind:start_ind + num_envs_per_batch].to(dev)
            take = lambda t: ops.take_envs(t, idx, ident)  # noqa: E731
            names = self.observations if sensors is None else [s for s in self.observations if s in sensors]
            observations_batch = {s: take(self.observations[s][:-1]) for s in names}
            out = (
                observations_batch,
                take(self.prev_pred_monoFromMem[1:]),
                take(self.prev_pred_monoFromMem[:-1]),
                take(self.masks[:-1]),
            )
            yield out + (idx,) if with_perm else out
