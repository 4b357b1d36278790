"""PPO / DD-PPO on MI355X: drop-in for audio_separation/rl/ppo/ppo.py (PPO :11-272, DecentralizedDistributedMixin :275-319).

Same constructor arguments, attributes (optimizer_pol, optimizer_sep, clip_param ...) and methods (update_pol, update_sep,
get_advantages, load_pretrained_passive_separators, init_distributed).  Mechanism:
  * every forward/backward value is produced by HIP kernels (m2h.functional); torch.autograd only orders them;
  * the clipped-surrogate / value / entropy losses and their gradients are ONE kernel (m2h_ppo_loss);
  * the two optimizers are FlatAdam: grad-norm clip + Adam in two launches over a flat buffer, no host sync;
  * DD-PPO: instead of DDP's bucketed reducer the flat gradient buffer is all-reduced (sum, RCCL) once per backward and
    averaged inside the optimizer step (same update as ppo.py:313-319 + DDP: grads averaged before clip and step);
    parameters are broadcast from rank 0 at init_distributed; the advantage normalisation uses the two scalar all-reduces
    of ddppo_utils.py:168-190;
  * update_sep (D13 of SURVEY): the frozen, eval-mode separators are deterministic per observation, so their outputs are
    computed once per stored buffer generation and re-used by the 4 epochs x 6 sub-updates that the reference recomputes
    (24x fewer U-Net passes, identical numbers); ``cache_separator_outputs=False`` restores the reference schedule.
  * loss scalars are accumulated on the device and read back once per update instead of 3 ``.item()`` per minibatch.
"""
import torch
import torch.nn as nn

from ... import functional as MF
from ... import graphs
from ... import ops
from ...optim import FlatAdam
from . import ddppo_utils

EPS_PPO = 1e-5


class PPO(nn.Module):
    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, bin_separation_loss_coef,
                 mono_conversion_loss_coef, entropy_coef, lr_pol=None, lr_sep=None, eps=None, max_grad_norm=None,
                 freeze_passive_separators=False, use_clipped_value_loss=True, use_normalized_advantage=True,
                 cache_separator_outputs=True, overlap_grad_reduce=None, use_hip_graphs=False, bucketed_grad_reduce=None):
        super().__init__()
        self.actor_critic = actor_critic
        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch
        self.value_loss_coef = value_loss_coef
        self.bin_separation_loss_coef = bin_separation_loss_coef
        self.mono_conversion_loss_coef = mono_conversion_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.use_normalized_advantage = use_normalized_advantage
        self.freeze_passive_separators = freeze_passive_separators
        self.cache_separator_outputs = cache_separator_outputs
        ac = actor_critic
        pol_params = list(ac.pol_net.parameters()) + list(ac.action_dist.parameters()) + list(ac.critic.parameters())
        sep_params = list(ac.binSep_enc.parameters()) + list(ac.binSep_dec.parameters()) + list(ac.bin2mono_enc.parameters()) + \
            list(ac.bin2mono_dec.parameters()) + list(ac.acoustic_mem.parameters())
        # flat buffers are built lazily at the first step, over the parameters that require grad THEN (the trainer freezes
        # the separators after constructing the agent, ppo_trainer.py:637-641)
        self.optimizer_pol = FlatAdam(pol_params, lr=lr_pol, eps=eps)
        self.optimizer_sep = FlatAdam(sep_params, lr=lr_sep, eps=eps)
        self.device = next(actor_critic.parameters()).device
        self._world = 1
        self._distributed = False
        self._sep_cache = None
        # overlap_grad_reduce: None = on when distributed (init_distributed), True = also at world size 1 (the side-stream
        # schedule without the collective; tests), False = the synchronous order of the reference
        self.overlap_grad_reduce = overlap_grad_reduce
        # bucketed_grad_reduce: the policy gradient's all-reduce in two buckets, the first (recurrent encoder + heads) under the encoders'
        # backward (ddppo_utils.GradReduceStep.early).  None = on when distributed, True = also at world size 1 (the schedule without the
        # collective; tests), False = one flat all-reduce behind the whole backward
        self.bucketed_grad_reduce = bucketed_grad_reduce
        # use_hip_graphs: replay one epoch of update_pol (forward, losses, backward) from a HIP graph (same kernels and values;
        # the epoch is ~400 small launches behind ~5 ms of Python and autograd dispatch)
        self.use_hip_graphs = use_hip_graphs
        self._pol_graph = None
        self._pol_updates = 0
        self._reducers = {"pol": ddppo_utils.GradReduceStep(), "mem": ddppo_utils.GradReduceStep()}
        actor_critic._param_fences = self._reducers  # readers of the parameters fence on these (policy.py)

    def load_pretrained_passive_separators(self, state_dict):
        ac = self.actor_critic
        for mod_name in ("binSep_enc", "binSep_dec", "bin2mono_enc", "bin2mono_dec"):
            mod = getattr(ac, mod_name)
            sd = mod.state_dict()
            for name in sd:
                sd[name].copy_(state_dict["actor_critic." + mod_name + "." + name])

    def forward(self, *x):
        raise NotImplementedError

    # ------------------------------------------------------------------ advantages
    def get_advantages(self, rollouts_pol):
        if not self.use_normalized_advantage:
            return ops.advantages(rollouts_pol.returns, rollouts_pol.value_preds, 0)[0]
        if self._distributed:   # DDPPO after init_distributed -- at world size 1 too (reference :311: get_advantages is re-bound)
            return self._get_advantages_distributed(rollouts_pol)
        return ops.advantages(rollouts_pol.returns, rollouts_pol.value_preds, 1, EPS_PPO)[0]

    def _get_advantages_distributed(self, rollouts_pol):
        adv, stats = ops.advantages(rollouts_pol.returns, rollouts_pol.value_preds, 2)
        return ddppo_utils.normalize_advantages_distributed(adv, stats[0:1], ops.adv_sqdiff, ops.adv_apply, EPS_PPO)

    # ------------------------------------------------------------------ distributed
    def init_distributed(self, find_unused_params: bool = True) -> None:
        """Broadcast rank 0's parameters/buffers and switch gradient reduction on (reference :286-311)."""
        self._world = ddppo_utils.world_size()
        self._distributed = True
        self.find_unused_params = find_unused_params
        ddppo_utils.broadcast_parameters(list(self.actor_critic.parameters()) + list(self.actor_critic.buffers()))

    def _overlap(self):
        return self._world > 1 if self.overlap_grad_reduce is None else bool(self.overlap_grad_reduce)

    def _bucketed(self):
        return self._world > 1 if self.bucketed_grad_reduce is None else bool(self.bucketed_grad_reduce)

    def _tail_bucket_params(self):
        """The parameters whose gradients are complete when the backward reaches the encoders' features: the recurrent state encoder
        and the two heads -- the tail of the policy optimizer's flat buffer (the encoders come first in pol_net.parameters())."""
        ac = self.actor_critic
        return [q for q in list(ac.pol_net.state_encoder.parameters()) + list(ac.action_dist.parameters()) + list(ac.critic.parameters())
                if q.requires_grad]

    def _reduce_and_step(self, group, opt, last, reduced_tail=None):
        """One flat sum all-reduce (RCCL) + clip + Adam.  The last mini-batch of an update is deferred onto the side stream
        when overlap is on: nothing in the rest of the update reads these parameters (ddppo_utils.GradReduceStep).
        reduced_tail: parameters whose bucket ``GradReduceStep.early`` has already all-reduced (the tail of the flat buffer): the
        collective here covers the rest of the buffer only."""
        flat = opt.grad_buffer()
        if reduced_tail is not None:
            b0, b1, _idx = opt.param_range(reduced_tail)
            if b1 != flat.numel():
                raise RuntimeError("bucketed gradient reduction: the early bucket must be the tail of the flat buffer")
            flat = flat[:b0]
        self._reducers[group].submit(
            flat, lambda gscale: opt.step(max_grad_norm=self.max_grad_norm, grad_scale=gscale),
            defer=last and self._overlap())

    def synchronize_updates(self):
        """Orders the current stream after every pending optimizer step (before reading parameters outside the policy's
        own methods: checkpoints, tests)."""
        for r in self._reducers.values():
            r.fence()

    # ------------------------------------------------------------------ policy update (reference :82-177)
    def _pol_epoch(self, sample, clip, acc, prepared=None, split=False):
        """Forward, losses and backward of one mini-batch (reference :94-163); the optimizer step follows in the caller.
        split: the backward stops at the encoders' features (gradients of the recurrent encoder and the heads complete) and returns
        them; ``_pol_epoch_rest`` runs the encoders' backward (the bucketed gradient reduction goes in between)."""
        (obs_batch, h_batch, pm_batch, mono_batch, mem_batch, value_preds_batch, return_batch, adv_targ, actions_batch,
         old_logp_batch, masks_batch) = sample
        net = self.actor_critic.pol_net
        net.keep_encoder_features = bool(split)
        try:
            values, logp, ent_rows, _ = self.actor_critic.evaluate_rows(
                obs_batch, h_batch, masks_batch, actions_batch, pred_binSepMasks=pm_batch, pred_mono=mono_batch,
                pred_monoFromMem=mem_batch, prepared=prepared)
        finally:
            net.keep_encoder_features = False
        feats, net.encoder_features = net.encoder_features, None   # (the concatenated encoder outputs of THIS forward, non-leaf; nothing is kept on the module)
        self.optimizer_pol.zero_grad()
        total_loss, stats = MF.PPOLoss.apply(values, logp, ent_rows, value_preds_batch, return_batch, adv_targ, old_logp_batch,
                                             clip, float(self.value_loss_coef), float(self.entropy_coef),
                                             bool(self.use_clipped_value_loss))
        if split:
            torch.autograd.backward(total_loss, MF.unit_grad(total_loss.device), inputs=[feats] + self._tail_bucket_params())
        else:
            total_loss.backward(MF.unit_grad(total_loss.device))
        acc += stats
        return feats

    def _pol_epoch_rest(self, feats):
        """The encoders' backward from the features' gradient (second half of a split ``_pol_epoch``)."""
        g = feats.grad
        feats.grad = None
        torch.autograd.backward(feats, g)

    def update_pol(self, rollouts_pol, as_tensor=False):
        """as_tensor: return the three mean losses as a device tensor instead of python floats (no host synchronisation at the end of the
        update: the trainer's next rollout steps are enqueued under the last epoch's optimizer step)."""
        advantages = self.get_advantages(rollouts_pol)
        self._pol_updates += 1
        if (self.use_hip_graphs and self._pol_updates > 1 and self.num_mini_batch == 1 and not ops.timing_enabled()
                and getattr(rollouts_pol, "full_batch_views", False)):
            return self._update_pol_graph(rollouts_pol, advantages, as_tensor)  # (the first update runs kernel by kernel: warm-up)
        self._pol_graph = None  # an eager epoch re-binds p.grad to fresh tensors: a graph captured earlier would write stale ones
        acc = torch.zeros(4, device=self.device)
        for _e in range(self.ppo_epoch):
            for _mb, sample in enumerate(rollouts_pol.recurrent_generator(advantages, self.num_mini_batch)):
                tail = None
                if self._bucketed():
                    tail = self._tail_bucket_params()
                    feats = self._pol_epoch(sample, float(self.clip_param), acc, split=True)
                    self._reducers["pol"].early(self.optimizer_pol.grad_bucket(tail))
                    self._pol_epoch_rest(feats)
                else:
                    self._pol_epoch(sample, float(self.clip_param), acc)
                self._reduce_and_step("pol", self.optimizer_pol,  # before_step_pol + step
                                      last=_e == self.ppo_epoch - 1 and _mb == self.num_mini_batch - 1, reduced_tail=tail)
        num_updates = self.ppo_epoch * self.num_mini_batch
        if as_tensor:
            return (acc / num_updates)[:3]
        v, a, h, _ = (acc / num_updates).tolist()  # the only host read of the update
        return v, a, h

    def _update_pol_graph(self, rollouts_pol, advantages, as_tensor=False):
        """update_pol with each epoch's forward + losses + backward replayed from one HIP graph.  With one mini-batch the batch
        is the whole storage read in place (rollout_storage.py), so every address the epoch touches is fixed: storages, flat
        parameter / gradient buffers, and three static inputs -- the advantages, the clip range (device scalar, it decays per
        update) and the loss accumulator.  Outside the graph stay: the advantage statistics (collectives), the CPU randperm
        (drawn as in the kernel-by-kernel path, so the generator streams stay aligned), the in-place re-pack of the conv
        weights after each optimizer step, the gradient all-reduce and the optimizer step."""
        gs = self._pol_graph
        self.optimizer_pol.build()
        sig = (id(rollouts_pol), rollouts_pol.rewards.data_ptr(), tuple(advantages.shape), float(self.value_loss_coef),
               float(self.entropy_coef), bool(self.use_clipped_value_loss), ops.math_mode(), self._bucketed(),
               tuple(p.data_ptr() for p in self.optimizer_pol.param_groups[0]["params"]))
        if gs is None or gs.sig != sig:
            from types import SimpleNamespace
            gs = self._pol_graph = SimpleNamespace(sig=sig, graph=None, graph_rest=None, feats=None, bucketed=self._bucketed(), forked=False,
                                                   prepared=None, adv=torch.empty_like(advantages),
                                                   clip=torch.zeros(1, device=self.device), acc=torch.zeros(4, device=self.device))
        gs.adv.copy_(advantages)
        gs.clip.fill_(float(self.clip_param))
        gs.acc.zero_()
        num_envs = rollouts_pol.rewards.size(1)
        # the encoders' input glue (rgb-d scaling, the two audio inputs' slicing) depends on the stored batch alone: once per update, into
        # the buffers the epoch's graph reads (it was the first kernel of each of the graph's three branches, four times per update)
        cpu_rng = torch.get_rng_state()
        s0 = next(iter(rollouts_pol.recurrent_generator(gs.adv, 1)))     # (the batch is the storage in place: views; its randperm draw is undone)
        torch.set_rng_state(cpu_rng)
        with torch.no_grad():
            gs.prepared = self.actor_critic.pol_net.prepare_inputs(s0[0], s0[2], s0[3], s0[4], out=gs.prepared)
        for _e in range(self.ppo_epoch):
            self._reducers["pol"].fence()   # the graph holds no fence: order it after a pending optimizer step here
            # conv weights re-packed in place after the previous step (the rollout's fused audio pair is rebuilt lazily, by its next user).
            # AcousticMem's packs are not this update's business -- and not its right: the trainer may be running update_sep on a second
            # stream at this moment (ppo_trainer.py, the cycle's tail), where that module re-packs its own weights in its own order.
            # (Restricting this to the policy's own memos -- the frozen separators' look stale after every step too, their keys carry the global
            # parameter epoch -- was measured in round 5: update_pol 42.1 / 42.4 / 42.5 against 41.6 / 42.6 / 42.3 ms per cycle: the epoch is
            # host-bound around this point, the 108 us pack launch hides in that.)
            mem_ids = {id(m) for m in MF.memos_of(self.actor_critic.acoustic_mem)}
            MF.refresh_pack_memos(hooks=False, only=[m for m in list(MF._pack_memos) if id(m) not in mem_ids])
            if gs.graph is None:
                cpu_rng = torch.get_rng_state()  # capture executes the python once without running kernels: no RNG side effect
                g = torch.cuda.CUDAGraph()
                if gs.bucketed:
                    # two graphs: forward + losses + the backward down to the encoders' features | the encoders' backward.  Between their
                    # replays the first bucket (recurrent encoder + heads) goes to the side stream for its all-reduce.
                    with graphs.capture(g):
                        gs.feats = self._pol_epoch(next(iter(rollouts_pol.recurrent_generator(gs.adv, 1))), gs.clip, gs.acc, prepared=gs.prepared,
                                                   split=True)
                    g2 = torch.cuda.CUDAGraph()
                    with graphs.capture(g2, pool=g.pool()):
                        self._pol_epoch_rest(gs.feats)
                    gs.graph_rest = g2
                else:
                    with graphs.capture(g):
                        self._pol_epoch(next(iter(rollouts_pol.recurrent_generator(gs.adv, 1))), gs.clip, gs.acc, prepared=gs.prepared)
                torch.set_rng_state(cpu_rng)
                gs.graph, gs.forked = g, graphs.parallel_branches
            torch.randperm(num_envs)        # recurrent_generator's draw (:197); the batch itself is the storage in place
            if gs.forked:
                # the epoch's graph has parallel branches (the three encoders, m2h/graphs.py): it is launched onto a drained stream --
                # queued behind the rollout's replays its side queues' parked barrier packets slow every kernel boundary ahead of it
                torch.cuda.current_stream().synchronize()
            graphs.replay(gs.graph)
            tail = None
            if gs.bucketed:
                tail = self._tail_bucket_params()
                self._reducers["pol"].early(self.optimizer_pol.grad_bucket(tail))
                graphs.replay(gs.graph_rest)
            self._reduce_and_step("pol", self.optimizer_pol, last=_e == self.ppo_epoch - 1, reduced_tail=tail)
        if as_tensor:
            return (gs.acc / self.ppo_epoch)[:3]   # (a new tensor: the next update zeroes gs.acc)
        v, a, h, _ = (gs.acc / self.ppo_epoch).tolist()
        return v, a, h

    # ------------------------------------------------------------------ separator (acoustic memory) update (reference :179-246)
    def _separator_outputs(self, rollouts_sep):
        """pred_binSepMasks / pred_mono for every stored (t, env), storage order [T, N]."""
        key = (id(rollouts_sep), getattr(rollouts_sep, "generation", None), rollouts_sep.observations["mixed_bin_audio_mag"].data_ptr())
        if self.cache_separator_outputs and self._sep_cache is not None and self._sep_cache[0] == key and key[1] is not None:
            return self._sep_cache[1]
        stored = getattr(rollouts_sep, "stored_separator_outputs", None)
        stored = stored() if (stored is not None and self.cache_separator_outputs) else None
        if stored is not None:
            # the trainer's rollout steps left every stored observation's outputs beside it (ppo_trainer.py): nothing to compute
            self._sep_cache = (key, stored)
            return stored
        mix = rollouts_sep.observations["mixed_bin_audio_mag"][:-1]
        T, N = mix.shape[0], mix.shape[1]
        tcl = rollouts_sep.observations["target_class"][:-1]
        prev = self._sep_cache
        if (self.cache_separator_outputs and prev is not None and key[1] is not None and prev[0] == (key[0], key[1] - 1, key[2])
                and getattr(rollouts_sep, "row0_only_since", None) == key[1] - 1):
            # after_update() between two sub-updates copied the last stored observation into row 0 and touched nothing else
            # (rollout_storage.py): refresh the N cached samples of that row instead of the T*N of the whole buffer
            with torch.no_grad():
                pm0 = self.actor_critic.get_binSepMasks({"mixed_bin_audio_mag": mix[0], "target_class": tcl[0]})
                mono0 = self.actor_critic.convert_bin2mono(pm0, mixed_audio=mix[0])
            prev[1][0][0].copy_(pm0)
            prev[1][1][0].copy_(mono0)
            val = (prev[1][0], prev[1][1])  # (the cached logging losses are whole-buffer means: dropped, recomputed once)
            self._sep_cache = (key, val)
            return val
        obs = {"mixed_bin_audio_mag": mix.reshape(T * N, *mix.shape[2:]), "target_class": tcl.reshape(T * N, -1)}
        with torch.no_grad():
            pm = self.actor_critic.get_binSepMasks(obs)
            mono = self.actor_critic.convert_bin2mono(pm, mixed_audio=obs["mixed_bin_audio_mag"])
        val = (pm.view(T, N, *pm.shape[1:]), mono.view(T, N, *mono.shape[1:]))
        self._sep_cache = (key, val)
        return val

    def _sep_cache_add_losses(self, bin_loss, mono_loss):
        key, val = self._sep_cache
        val = (val[0], val[1], (bin_loss.detach(), mono_loss.detach()))
        self._sep_cache = (key, val)
        return val

    _sep_graph = None
    _sep_updates = 0

    def _sep_graph_state(self, pred_mono, prev_mem, masks, gt_mono):
        """Static state of the update_sep epoch's graph: the memory's sliced input lives in a buffer of its own (written once per update),
        everything else the epoch reads is a view of the separator storage or lives in the optimizer's flat buffers."""
        self.optimizer_sep.build()
        sig = (pred_mono.data_ptr(), prev_mem.data_ptr(), masks.data_ptr(), gt_mono.data_ptr(), tuple(pred_mono.shape), ops.math_mode(),
               tuple(p.data_ptr() for p in self.optimizer_sep.param_groups[0]["params"]))
        gs = self._sep_graph
        if gs is None or gs.sig != sig:
            from types import SimpleNamespace
            B, Fq, T, _ = pred_mono.shape
            gs = self._sep_graph = SimpleNamespace(sig=sig, graph=None, loss=None,
                                                   sliced=torch.empty((B, Fq // 16, T, 32), device=pred_mono.device, dtype=torch.float32),
                                                   gt_plane=torch.empty((B, Fq, T, 1), device=pred_mono.device, dtype=torch.float32),
                                                   memos=MF.memos_of(self.actor_critic.acoustic_mem))
        return gs

    def _sep_epoch_graph(self, gs, pred_mono, prev_mem, masks, gt_mono):
        self._reducers["mem"].fence()       # the graph holds no fence: order it after a pending optimizer step here
        MF.refresh_pack_memos(hooks=False, only=gs.memos)   # the memory's conv weights re-packed in place after the previous step
        if gs.graph is None:
            g = torch.cuda.CUDAGraph()
            with graphs.capture(g):
                self.optimizer_sep.zero_grad()
                loss = self.actor_critic.monoFromMem_l1_masked(pred_mono, prev_mem, masks, gt_mono, 0, sliced=gs.sliced)
                loss.backward(MF.unit_grad(loss.device))
                gs.loss = loss.detach()
            gs.graph = g
        graphs.replay(gs.graph)
        return gs.loss

    def update_sep(self, rollouts_sep, as_tensor=False):
        """as_tensor: return the three mean losses as a device tensor instead of python floats (no host synchronisation: the
        trainer enqueues the cycle's separator updates on a second stream and reads the losses after the join)."""
        self._sep_updates += 1
        acc = torch.zeros(3, device=self.device)
        sep_frozen = not any(p.requires_grad for m in (self.actor_critic.binSep_enc, self.actor_critic.binSep_dec,
                                                        self.actor_critic.bin2mono_enc, self.actor_critic.bin2mono_dec)
                             for p in m.parameters())
        # separators a caller left unfrozen (the reference's own trainer never does: ppo_trainer.py:557-577, :637-638): update_sep
        # still runs them under no_grad, in whatever mode the modules are in, and back-propagates only the memory's loss (:184-195,
        # :226); every pass of the reference is then made (no cached outputs: train-mode BatchNorm would move its statistics)
        cached = self._separator_outputs(rollouts_sep) if (self.cache_separator_outputs and sep_frozen) else None
        sliced = None   # the memory's sliced + concatenated + masked input: the same tensor in every epoch of a full-batch update
        for _e in range(self.ppo_epoch):
            needed = ("mixed_bin_audio_mag", "gt_mono_comps", "gt_bin_comps", "target_class")  # what this update reads
            gen = rollouts_sep.recurrent_generator(self.num_mini_batch, with_perm=True, sensors=needed)
            for _mb, sample in enumerate(gen):
                obs_batch, _mem_batch, prev_mem_batch, masks_batch, idx = sample
                if cached is not None:
                    pred_binSepMasks = ops.take_envs(cached[0], idx, idx is None)
                    pred_mono = ops.take_envs(cached[1], idx, idx is None)
                else:
                    with torch.no_grad():  # reference :184-195
                        pred_binSepMasks = self.actor_critic.get_binSepMasks(obs_batch)
                        pred_mono = self.actor_critic.convert_bin2mono(pred_binSepMasks.detach(),
                                                                       mixed_audio=obs_batch["mixed_bin_audio_mag"])
                gt_mono = obs_batch["gt_mono_comps"]
                graphed = (cached is not None and idx is None and self.use_hip_graphs and self._sep_updates > 1 and not ops.timing_enabled()
                           and getattr(rollouts_sep, "full_batch_views", False) and pred_mono.shape[1] == 512)
                if graphed:
                    # forward + loss + backward of the epoch replayed from a HIP graph (one chain; same kernels, same values): the 20 launches
                    # of an epoch no longer wait for the host one by one
                    gs = self._sep_graph_state(pred_mono, prev_mem_batch, masks_batch, gt_mono)
                    if sliced is None:
                        # the sliced input is a function of the stored batch alone: the same tensor for every update on one generation of the
                        # storage (the cycle's updates 2-6: after_update() changes nothing once row 0 holds the last observation), as the
                        # cached separator outputs are (a 330 MB pass, 123 us, per update otherwise)
                        skey = (id(rollouts_sep), getattr(rollouts_sep, "generation", None), rollouts_sep.prev_pred_monoFromMem.data_ptr())
                        if skey[1] is None or getattr(gs, "sliced_key", None) != skey:
                            with torch.no_grad():
                                self.actor_critic.acoustic_mem.slice_inputs(pred_mono, prev_mem_batch, masks_batch, out=gs.sliced)
                                # the loss's target, gt_mono_comps[..., 0] (:210-216), as a plane of its own: the stored tensor interleaves four
                                # components per bin, so the loss kernel of every epoch fetched 440 MB of lines for 110 MB of magnitudes
                                # (bench.py update_sep kernels: 126 us, the epoch's worst against its bytes).  Same floats, read once per
                                # storage generation here instead of 24 x per cycle there.
                                gs.gt_plane.copy_(gt_mono[..., 0:1])
                            gs.sliced_key = skey
                        sliced = gs.sliced
                    monoFromMem_loss = self._sep_epoch_graph(gs, pred_mono, prev_mem_batch, masks_batch, gs.gt_plane)
                elif cached is not None and idx is None:
                    if sliced is None:
                        with torch.no_grad():
                            sliced = self.actor_critic.acoustic_mem.slice_inputs(pred_mono, prev_mem_batch, masks_batch)
                            gt_plane = gt_mono[..., 0:1].contiguous()   # (as the graphed epoch: the loss kernel reads the target as a plane)
                    # the memory's output is only this loss's operand here: it stays in its conv's layout (no de-slice / re-slice round trips)
                    monoFromMem_loss = self.actor_critic.monoFromMem_l1_masked(pred_mono, prev_mem_batch, masks_batch, gt_plane, 0, sliced=sliced)
                else:
                    pred_monoFromMem = self.actor_critic.get_monoFromMem_masked(pred_mono, prev_mem_batch, masks_batch)
                    monoFromMem_loss = MF.l1_loss(pred_monoFromMem, gt_mono, 0)      # gt_mono_comps[..., 0::2][..., :1]
                if cached is not None and idx is None and len(cached) > 2:
                    bin_loss, mono_loss = cached[2]  # logging losses of the frozen separators: same buffer, same numbers
                else:
                    with torch.no_grad():
                        mono_loss = MF.l1_loss(pred_mono, gt_mono, 0)
                        bin_loss = ops.bin_l1_loss(obs_batch["mixed_bin_audio_mag"], pred_binSepMasks, obs_batch["gt_bin_comps"])
                    if cached is not None and idx is None:
                        # the two losses that are only logged (:219-224) are functions of the stored observations and the cached
                        # separator outputs alone: computed once per buffer generation like those outputs
                        cached = self._sep_cache_add_losses(bin_loss, mono_loss)
                if not graphed:
                    self._sep_graph = None  # an eager epoch re-binds p.grad: a graph captured earlier would gather stale gradients (as update_pol)
                    self.optimizer_sep.zero_grad()
                    monoFromMem_loss.backward(MF.unit_grad(monoFromMem_loss.device))     # total_loss = monoFromMem_loss (:226)
                self._reduce_and_step("mem", self.optimizer_sep, last=_e == self.ppo_epoch - 1 and _mb == self.num_mini_batch - 1)
                acc += torch.stack((bin_loss, mono_loss, monoFromMem_loss.detach()))
        num_updates = self.ppo_epoch * self.num_mini_batch
        if as_tensor:
            return acc / num_updates
        b, m, mm = (acc / num_updates).tolist()
        return b, m, mm


class DecentralizedDistributedMixin:
    """Kept for API parity (reference :275-319): the distributed behaviour lives in PPO itself and is switched on by
    init_distributed()."""


class DDPPO(DecentralizedDistributedMixin, PPO):
    pass
