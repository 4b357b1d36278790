"""Move2Hear RL policy on MI355X: drop-in for audio_separation/rl/ppo/policy.py.

Same class names (CriticHead, PolicyNet, PassiveSepEnc/Dec, Policy, Move2HearPolicy), constructor arguments, method
signatures/return conventions (:183-273) and state_dict keys (158 entries, checked against the reference).  Every
arithmetic step -- forward and backward -- runs in libm2h.so (m2h.functional wraps the kernels as autograd Functions); the
frozen separator U-Nets are inference-only, exactly how the RL trainer uses them (ppo_trainer.py:557-577).
"""
import abc

import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops
from ...common.utils import CategoricalNet, CustomFixedCategorical
from ...pretrain.passive.policy import PassiveSepDec, PassiveSepEnc  # identical wrappers (reference :121-156)
from ..models.audio_cnn import AudioCNN, FusedAudioPair
from ..models.separator_cnn import unet_forward
from ..models.memory_nets import AcousticMem
from ..models.rnn_state_encoder import RNNStateEncoder
from ..models.visual_cnn import VisualCNN


class CriticHead(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.fc = nn.Linear(input_size, 1)
        nn.init.orthogonal_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, x):
        return ops.linear(x.contiguous(), self.fc.weight.detach(), self.fc.bias.detach(), name="critic")


class Net(nn.Module, metaclass=abc.ABCMeta):
    @abc.abstractmethod
    def forward(self, observations, rnn_hidden_states, prev_actions, masks):
        pass

    @property
    @abc.abstractmethod
    def output_size(self):
        pass

    @property
    @abc.abstractmethod
    def num_recurrent_layers(self):
        pass

    @property
    @abc.abstractmethod
    def is_blind(self):
        pass


class PolicyNet(Net):
    r"""Visual + two audio encoders -> concat -> GRU (reference :46-118)."""

    def __init__(self, observation_space, hidden_size, goal_sensor_uuid, extra_rgb=False, extra_depth=False, world_rank=0):
        super().__init__()
        assert 'mixed_bin_audio_mag' in observation_space.spaces
        self.goal_sensor_uuid = goal_sensor_uuid
        self._hidden_size = hidden_size
        self.visual_encoder = VisualCNN(observation_space, hidden_size, extra_rgb, extra_depth)
        self.bin_encoder = AudioCNN(observation_space, hidden_size)
        self.monoNmonoFromMem_encoder = AudioCNN(observation_space, hidden_size, encode_monoNmonoFromMem=True)
        rnn_input_size = 3 * self._hidden_size
        self.state_encoder = RNNStateEncoder(rnn_input_size, self._hidden_size)
        self._audio_pair = FusedAudioPair(self.bin_encoder, self.monoNmonoFromMem_encoder)   # no parameters of its own (rollout fast path)
        self.encoder_features = None
        self.keep_encoder_features = False

    @property
    def is_blind(self):
        return False

    @property
    def output_size(self):
        return self._hidden_size

    @property
    def num_recurrent_layers(self):
        return self.state_encoder.num_recurrent_layers

    def prepare_inputs(self, observations, pred_binSepMasks, pred_mono, pred_monoFromMem, out=None):
        """The three encoders' NHWC inputs (visual: rgb-d / 255; binaural: log1p(clamp0(mix * masks)) sliced; mono: mono | memory sliced;
        :98-103 with visual_cnn.py:135-140, audio_cnn.py:118-128): functions of the stored batch alone, so update_pol makes them once per
        update instead of once per epoch.  out: a previous result to overwrite in place (the epoch's HIP graph reads these by address)."""
        o = out if out is not None else (None, None, None)
        xv = self.visual_encoder.prepare(observations, out=o[0])
        xa = ops.slice_concat_input(observations["mixed_bin_audio_mag"].contiguous(), mul=pred_binSepMasks.contiguous(), op=1, out=o[1])
        xb = ops.slice_concat_input(pred_mono.contiguous(), pred_monoFromMem.contiguous(), op=2, out=o[2])
        return xv, xa, xb

    def forward(self, observations, rnn_hidden_states, masks, pred_binSepMasks=None, pred_mono=None, pred_monoFromMem=None, prepared=None):
        from ... import graphs
        if self._audio_pair.usable(pred_mono, pred_monoFromMem) and pred_mono.shape[0] < 64:
            # rollout step (no gradients, 14 envs): the two audio encoders as one chain of block-diagonal layers (audio_cnn.FusedAudioPair)
            xa = ops.slice_concat_input(observations["mixed_bin_audio_mag"].contiguous(), mul=pred_binSepMasks.contiguous(), op=1)
            xb = ops.slice_concat_input(pred_mono.contiguous(), pred_monoFromMem.contiguous(), op=2)
            # the three encoders write their features side by side into one matrix (the cat of :103 is never a copy)
            hs = self._hidden_size
            x1 = torch.empty((pred_mono.shape[0], 3 * hs), device=pred_mono.device)
            self._audio_pair.encode(xa, xb, out=x1[:, hs:])
            self.visual_encoder(observations, out=x1[:, :hs])
            x2, rnn_hidden_states_new = self.state_encoder(x1, rnn_hidden_states, masks)
            return x2, rnn_hidden_states_new
        # The three encoders are independent kernel chains: at update batches, while a HIP graph is being captured (update_pol's
        # epoch, ppo.py), graphs.run_parallel puts them on three streams = three branches of the graph (forward and backward);
        # sequential otherwise (m2h/graphs.py has the measurements and the launch rule that goes with it).
        if prepared is not None:      # update_pol's epochs: the inputs' glue was made once for the update (prepare_inputs)
            xv, xa, xb = prepared
            fns = [lambda: self.visual_encoder(observations, x=xv), lambda: self.bin_encoder.encode(xa),
                   lambda: self.monoNmonoFromMem_encoder.encode(xb)]
        else:
            fns = [lambda: self.visual_encoder(observations),
                   lambda: self.bin_encoder(observations, pred_binSepMasks=pred_binSepMasks),
                   lambda: self.monoNmonoFromMem_encoder.forward_pair(pred_mono, pred_monoFromMem)]  # cat(dim=3) read in place
        x = graphs.run_parallel(pred_mono.device if pred_mono.shape[0] >= 64 else None, fns)
        x1 = torch.cat(x, dim=1)
        if self.keep_encoder_features:     # where a split backward stops (ppo.py, the bucketed gradient reduction): handed over once
            self.encoder_features = x1
        x2, rnn_hidden_states_new = self.state_encoder(x1, rnn_hidden_states, masks)
        # the reference asserts "not isnan(x2).any().item()" here (:116): a host sync per call; dropped.
        return x2, rnn_hidden_states_new


class Policy(nn.Module):
    r"""Full Move2Hear policy: separation + action-making (reference :159-273)."""

    def __init__(self, pol_net, dim_actions, binSep_enc, binSep_dec, bin2mono_enc, bin2mono_dec, acoustic_mem):
        super().__init__()
        self.dim_actions = dim_actions
        self.pol_net = pol_net
        self.action_dist = CategoricalNet(self.pol_net.output_size, self.dim_actions)
        self.critic = CriticHead(self.pol_net.output_size)
        self.binSep_enc = binSep_enc
        self.binSep_dec = binSep_dec
        self.bin2mono_enc = bin2mono_enc
        self.bin2mono_dec = bin2mono_dec
        self.acoustic_mem = acoustic_mem

    def forward(self):
        raise NotImplementedError

    # DD-PPO overlap (ppo.py, ddppo_utils.GradReduceStep): the optimizer step of a parameter group ("pol": pol_net + heads,
    # "mem": acoustic_mem) may still be running on the side stream; every method that reads the group waits on its fence.
    _param_fences = None

    def _fence(self, group):
        if self._param_fences is not None:
            self._param_fences[group].fence()

    def state_dict(self, *args, **kwargs):
        self._fence("pol")
        self._fence("mem")
        return super().state_dict(*args, **kwargs)

    def get_binSepMasks(self, observations):
        enc, dec = self.binSep_enc.passive_sep_encoder, self.binSep_dec.passive_sep_decoder
        if not enc.training and not dec.training and not ops.timing_enabled() and observations["mixed_bin_audio_mag"].shape[1] == 512:
            # eval mode (every RL call site): the whole U-Net is enqueued by one C call (m2h_unet_fwd)
            return unet_forward(enc, dec, observations["mixed_bin_audio_mag"], None, observations["target_class"])
        bottleneck_feats, lst_skip_feats = self.binSep_enc(observations)
        return self.binSep_dec(bottleneck_feats, lst_skip_feats)

    def convert_bin2mono(self, pred_binSepMasks, mixed_audio=None):
        enc, dec = self.bin2mono_enc.passive_sep_encoder, self.bin2mono_dec.passive_sep_decoder
        if not enc.training and not dec.training and not ops.timing_enabled() and mixed_audio.shape[1] == 512:
            return unet_forward(enc, dec, mixed_audio, pred_binSepMasks)
        bottleneck_feats, lst_skip_feats = self.bin2mono_enc(pred_binSepMasks, mixed_audio=mixed_audio)
        return self.bin2mono_dec(bottleneck_feats, lst_skip_feats)

    def get_monoFromMem(self, pred_mono, prev_pred_monoFromMem_masked):
        self._fence("mem")
        return self.acoustic_mem(pred_mono, prev_pred_monoFromMem_masked)

    def get_monoFromMem_masked(self, pred_mono, prev_pred_monoFromMem, masks, sliced=None):
        """get_monoFromMem with the not-done masking of the previous memory fused (ppo_trainer.py:310-319).
        sliced: AcousticMem.slice_inputs of the same arguments, when the caller holds it (update_sep's epochs share one)."""
        self._fence("mem")
        return self.acoustic_mem.forward_masked(pred_mono, prev_pred_monoFromMem, masks, sliced=sliced)

    def monoFromMem_l1_masked(self, pred_mono, prev_pred_monoFromMem, masks, gt_comps, off=0, sliced=None):
        """update_sep's loss (ppo.py:206-216): l1_loss(get_monoFromMem_masked(...), gt_comps[..., off]) with the memory's output left in
        its conv's layout (AcousticMem.l1_loss_masked)."""
        self._fence("mem")
        return self.acoustic_mem.l1_loss_masked(pred_mono, prev_pred_monoFromMem, masks, gt_comps, off, sliced=sliced)

    # action sampling (common/utils.py): None = noise from the device generator; a HostNoise = noise from the CPU default generator;
    # _rng_state = [seed, counter] on the device = noise drawn inside the heads kernel ("fused").  Neither is a parameter or a buffer of
    # the reference's state_dict (its keys are the checkpoint contract): _apply below moves them with the module, the trainer's
    # checkpoints carry the counter beside the state_dict (sampler_state / restore_sampler_state).
    _host_noise = None
    _rng_state = None
    _sampling_mode = "device"
    last_action_noise = None      # record_noise_rows: [rows, actions] Exp(1) noise of the last fused draw (what a parity test hands the oracle)

    def set_action_sampling(self, mode, seed=0, record_noise_rows=None):
        """"fused" (throughput default of the trainers): torch.multinomial's single draw, argmax(probs / Exp(1) noise), with the noise made
        inside the heads kernel by a counter-based generator (Philox4x32-10 keyed by `seed`, counter on the device): no generator launch
        in the rollout step.  "device": the same draw with torch's device generator supplying the noise (three more launches per step inside
        a HIP graph).  "cpu_generator": the noise taken from the CPU default generator, i.e. the actions of the reference PyTorch-CPU path
        for the same seed (common/utils.py:16-24 on a CPU policy).
        record_noise_rows (fused only): keep the noise of the last draw over that many rows in ``last_action_noise``.
        A second call in "fused" mode re-seeds the EXISTING state tensor in place (captured rollout graphs hold its address); whatever a
        call changes that a captured graph depends on shows in ``sampling_key()``, which the trainer compares before every replay."""
        from ...common.utils import HostNoise
        if mode not in ("fused", "device", "cpu_generator"):
            raise ValueError("action_sampling must be 'fused', 'device' or 'cpu_generator', got %r" % (mode,))
        dev = next(self.parameters()).device
        self._sampling_mode = mode
        self._host_noise = HostNoise(dev) if mode == "cpu_generator" else None
        if mode == "fused":
            state = torch.tensor([int(seed) & 0x7fffffffffffffff, 0], dtype=torch.int64, device=dev)
            if self._rng_state is not None and self._rng_state.device == state.device:
                self._rng_state.copy_(state)
            else:
                self._rng_state = state
        else:
            self._rng_state = None
        rows = int(record_noise_rows or 0)
        if mode == "fused" and rows > 0:
            if self.last_action_noise is None or tuple(self.last_action_noise.shape) != (rows, self.dim_actions) or self.last_action_noise.device != dev:
                self.last_action_noise = torch.zeros(rows, self.dim_actions, device=dev)
        else:
            self.last_action_noise = None

    def sampling_key(self):
        """What a captured rollout step depends on besides the weights: the mode and the addresses of the sampler's device state."""
        return (self._sampling_mode, None if self._rng_state is None else self._rng_state.data_ptr(),
                None if self.last_action_noise is None else self.last_action_noise.data_ptr())

    def sampler_state(self):
        """[seed, counter] of the fused sampler as python ints (a host read), or None: what a checkpoint keeps so that a resumed run does
        not replay the noise of the run's first steps."""
        return None if self._rng_state is None else [int(v) for v in self._rng_state.tolist()]

    def restore_sampler_state(self, state):
        if state is not None and self._rng_state is not None:
            self._rng_state.copy_(torch.tensor([int(state[0]), int(state[1])], dtype=torch.int64))

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)      # .to() / .cuda(): the sampler's device state moves with the parameters
        for name in ("_rng_state", "last_action_noise"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, fn(t))
        dev = next(self.parameters()).device
        if self._host_noise is not None and self._host_noise.device != dev:
            from ...common.utils import HostNoise
            self._host_noise = HostNoise(dev)
        return out

    def prepare_action_sampling(self, rows):
        """Allocates what a draw over ``rows`` rows needs outside any stream capture (cpu_generator mode: the pinned ring and the static
        device buffer a captured step reads)."""
        if self._host_noise is not None:
            self._host_noise.buffer(rows, self.dim_actions)

    def stage_action_noise(self, rows):
        """cpu_generator mode: draw the next step's noise and enqueue its upload (called ahead of a graph replay that samples)."""
        if self._host_noise is not None:
            self._host_noise.stage(rows, self.dim_actions)

    def _heads(self, feats, actions=None):
        a, c = self.action_dist.linear, self.critic.fc
        acts = actions.reshape(-1).contiguous() if actions is not None else None
        value, logp_act, ent, probs, logp_all = MF.PolicyHeads.apply(feats, a.weight, a.bias, c.weight, c.bias, acts)
        return value, CustomFixedCategorical(logp_all, probs, ent, self._host_noise), (logp_act if actions is not None else None)

    def evaluate_rows(self, observations, rnn_hidden_states_pol, masks, action, pred_binSepMasks=None, pred_mono=None,
                      pred_monoFromMem=None, prepared=None):
        """evaluate_actions with the per-row entropies (what the fused PPO-loss kernel consumes).
        prepared: ``pol_net.prepare_inputs`` of the same batch, when the caller holds it (update_pol's epochs share one)."""
        self._fence("pol")
        feats_pol, rnn_hidden_states_pol = self.pol_net(
            observations, rnn_hidden_states_pol, masks, pred_binSepMasks=pred_binSepMasks, pred_mono=pred_mono,
            pred_monoFromMem=pred_monoFromMem, prepared=prepared)
        value, dist, action_log_probs = self._heads(feats_pol, action)
        return value, action_log_probs, dist.entropy(), rnn_hidden_states_pol

    def act(self, observations, rnn_hidden_states_pol, masks, deterministic=False, pred_binSepMasks=None, pred_mono=None,
            pred_monoFromMem=None):
        self._fence("pol")
        feats_pol, rnn_hidden_states_pol = self.pol_net(
            observations, rnn_hidden_states_pol, masks, pred_binSepMasks=pred_binSepMasks.detach(),
            pred_mono=pred_mono.detach(), pred_monoFromMem=pred_monoFromMem.detach())
        if not torch.is_grad_enabled():
            # rollout / evaluation (no autograd): heads, the draw (or the mode) and its log-probability in ONE launch.  The noise is
            # the draw torch.multinomial makes -- Exp(1) from the device generator, or from the CPU default generator through the
            # pinned ring (set_action_sampling); same generator state in, same noise, same actions out as the per-op path below.
            a, c = self.action_dist.linear, self.critic.fc
            noise, rng = None, None
            M, A = feats_pol.shape[0], self.dim_actions
            if not deterministic:
                if self._host_noise is not None:
                    noise = self._host_noise.buffer(M, A) if torch.cuda.is_current_stream_capturing() else self._host_noise.stage(M, A)
                elif self._rng_state is not None and feats_pol.is_cuda:
                    rng = self._rng_state
                else:
                    noise = torch.empty((M, A), device=feats_pol.device).exponential_(1)
            value, _lpa, probs, _ent, action, action_log_probs = ops.policy_heads_act(
                feats_pol.contiguous(), a.weight.detach(), a.bias.detach(), c.weight.detach(), c.bias.detach(), noise, rng=rng,
                noise_out=self.last_action_noise if rng is not None and self.last_action_noise is not None and self.last_action_noise.shape[0] == M else None)
            if rng is not None and not torch.cuda.is_current_stream_capturing():
                rng[1:2] += M * A     # kernel-by-kernel steps: the counter moves on here; a captured step advances it in its last launch (ppo_trainer.py)
            return value, action, action_log_probs, rnn_hidden_states_pol, probs
        value, dist, _ = self._heads(feats_pol)
        action = dist.mode() if deterministic else dist.sample()
        action_log_probs = dist.log_probs(action)
        return value, action, action_log_probs, rnn_hidden_states_pol, dist.get_probs()

    def get_value(self, observations, rnn_hidden_states_pol, masks, pred_binSepMasks=None, pred_mono=None, pred_monoFromMem=None):
        self._fence("pol")
        feats_pol, _ = self.pol_net(
            observations, rnn_hidden_states_pol, masks, pred_binSepMasks=pred_binSepMasks.detach(),
            pred_mono=pred_mono.detach(), pred_monoFromMem=pred_monoFromMem.detach())
        value, _, _ = self._heads(feats_pol)
        return value

    def evaluate_actions(self, observations, rnn_hidden_states_pol, masks, action, pred_binSepMasks=None, pred_mono=None,
                         pred_monoFromMem=None):
        self._fence("pol")
        feats_pol, rnn_hidden_states_pol = self.pol_net(
            observations, rnn_hidden_states_pol, masks, pred_binSepMasks=pred_binSepMasks, pred_mono=pred_mono,
            pred_monoFromMem=pred_monoFromMem)
        value, dist, action_log_probs = self._heads(feats_pol, action)
        dist_entropy = dist.entropy().mean()
        return value, action_log_probs, dist_entropy, rnn_hidden_states_pol


class Move2HearPolicy(Policy):
    def __init__(self, observation_space, action_space, goal_sensor_uuid, hidden_size=512, extra_rgb=False, extra_depth=False,
                 use_ddppo=False, world_rank=0, use_smartnav_for_eval_pol_mix=False):
        pol_net = PolicyNet(observation_space=observation_space, hidden_size=hidden_size, goal_sensor_uuid=goal_sensor_uuid,
                            extra_rgb=extra_rgb, extra_depth=extra_depth, world_rank=world_rank)
        binSep_enc = PassiveSepEnc(observation_space=observation_space, world_rank=world_rank)
        binSep_dec = PassiveSepDec()
        bin2mono_enc = PassiveSepEnc(observation_space=observation_space, world_rank=world_rank, convert_bin2mono=True)
        bin2mono_dec = PassiveSepDec(convert_bin2mono=True)
        acoustic_mem = AcousticMem(use_ddppo=use_ddppo)
        super().__init__(pol_net, action_space.n, binSep_enc, binSep_dec, bin2mono_enc, bin2mono_dec, acoustic_mem)
