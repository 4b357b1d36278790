"""GPU: FusedAudioPair -- the policy's two audio encoders (rl/ppo/policy.py:65-66, :87-89; audio_cnn.py:50-75) as one chain of
block-diagonal layers on the rollout's no-grad fast path (m2h/rl/models/audio_cnn.py) -- against the two AudioCNN modules it
stands for: eagerly, after an optimizer step (FlatAdam updates parameters through raw pointers), after load_state_dict, inside a
replayed HIP graph with the weights refreshed between replays, for a deep copy of the policy, and next to ANOTHER model's graph
capture (a passive training step captured while a PPO policy is alive: the refresh hooks must neither raise nor rebuild there)."""
import copy

import numpy as np
import pytest
import torch

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _policy(seed, dev):
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.rl.ppo.policy import Move2HearPolicy
    pol = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    pol.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}, strict=True)
    return pol.to(dev)


def _inputs(n, seed, dev):
    obs = {k: torch.from_numpy(v).float().to(dev) for k, v in synthetic.make_rl_observations(n, seed).items()}
    g = torch.Generator().manual_seed(seed)
    pm = torch.randn(n, 512, 32, 2, generator=g).to(dev)
    mono, mem = torch.rand(n, 512, 32, 1, generator=g).to(dev), torch.rand(n, 512, 32, 1, generator=g).to(dev)
    return obs, pm, mono, mem


def _separate(net, obs, pm, mono, mem):
    """The two encoders as the reference runs them (policy.py:98-103): separate modules, separate kernels."""
    with torch.no_grad():
        fa = net.bin_encoder(obs, pred_binSepMasks=pm)
        fb = net.monoNmonoFromMem_encoder.forward_pair(mono, mem)
    return fa, fb


def _fused(net, obs, pm, mono, mem):
    from m2h import ops
    with torch.no_grad():
        xa = ops.slice_concat_input(obs["mixed_bin_audio_mag"].contiguous(), mul=pm.contiguous(), op=1)
        xb = ops.slice_concat_input(mono.contiguous(), mem.contiguous(), op=2)
        fa, fb = net._audio_pair.encode(xa, xb)
    return fa.clone(), fb.clone()


def _close(a, b, tol=2e-6):
    return O.rel_l1(a.cpu(), b.cpu()) < tol


def test_fused_pair_is_the_two_encoders_and_follows_their_weights():
    from m2h import functional as MF
    from m2h.optim import FlatAdam
    dev = _dev()
    pol = _policy(3, dev)
    net = pol.pol_net
    args = _inputs(14, 5, dev)
    sa, sb = _separate(net, *args)
    fa, fb = _fused(net, *args)
    assert _close(fa, sa) and _close(fb, sb)
    # an optimizer step through the flat buffers (no torch version counter moves): the fused copies follow
    params = [p for n, p in pol.named_parameters() if n.startswith("pol_net.")]
    opt = FlatAdam(params, lr=1e-2, eps=1e-5)
    opt.zero_grad()
    for p in params:
        p.grad = torch.randn_like(p) * 0.1
    opt.step(max_grad_norm=None)
    sa2, sb2 = _separate(net, *args)
    assert not _close(sa2, sa, 1e-4)                         # the step did move the encoders
    fa2, fb2 = _fused(net, *args)
    assert _close(fa2, sa2) and _close(fb2, sb2)
    # load_state_dict: new values at the same addresses
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 9).items()}
    pol.load_state_dict(sd, strict=True)
    MF.bump_param_epoch()
    sa3, sb3 = _separate(net, *args)
    fa3, fb3 = _fused(net, *args)
    assert _close(fa3, sa3) and _close(fb3, sb3) and not _close(sa3, sa2, 1e-4)
    # a deep copy owns a fused pair of its own, registered for refreshes, reading the COPY's weights
    twin = copy.deepcopy(pol)
    assert twin.pol_net._audio_pair is not net._audio_pair and twin.pol_net._audio_pair.a is twin.pol_net.bin_encoder
    assert twin.pol_net._audio_pair in list(MF._refresh_hooks)
    with torch.no_grad():
        for p in twin.pol_net.bin_encoder.parameters():
            p.mul_(1.5)
    ta, tb = _separate(twin.pol_net, *args)
    ga, gb = _fused(twin.pol_net, *args)
    assert _close(ga, ta) and _close(gb, tb) and not _close(ta, sa3, 1e-4)


def test_fused_pair_inside_a_replayed_graph_reads_refreshed_weights():
    """The rollout step's graph holds the block-diagonal tensors and their packs by address (ppo_trainer.py): after an optimizer
    step, functional.refresh_pack_memos() rebuilds them in place and the replay computes with the new weights."""
    from m2h import functional as MF
    from m2h import graphs, ops
    from m2h.optim import FlatAdam
    dev = _dev()
    pol = _policy(4, dev)
    net = pol.pol_net
    obs, pm, mono, mem = _inputs(14, 6, dev)
    params = [p for n, p in pol.named_parameters() if n.startswith("pol_net.")]
    opt = FlatAdam(params, lr=1e-2, eps=1e-5)
    opt.build()                                              # parameters move into the flat buffer before anything holds addresses
    MF.refresh_pack_memos()
    _fused(net, obs, pm, mono, mem)                          # warm-up: block-diagonal tensors and packs exist
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), graphs.capture(g):
        xa = ops.slice_concat_input(obs["mixed_bin_audio_mag"].contiguous(), mul=pm.contiguous(), op=1)
        xb = ops.slice_concat_input(mono.contiguous(), mem.contiguous(), op=2)
        fa, fb = net._audio_pair.encode(xa, xb)
    g.replay()
    sa, sb = _separate(net, obs, pm, mono, mem)
    assert _close(fa, sa) and _close(fb, sb)
    opt.zero_grad()
    for p in params:
        p.grad = torch.randn_like(p) * 0.1
    opt.step(max_grad_norm=None)
    MF.refresh_pack_memos()                                  # what the trainer does before the next replay
    g.replay()
    sa2, sb2 = _separate(net, obs, pm, mono, mem)
    assert not _close(sa2, sa, 1e-4) and _close(fa, sa2) and _close(fb, sb2)
    # a capture that would bake STALE block-diagonal tensors in is refused where they are used, not in the hooks
    opt.zero_grad()
    for p in params:
        p.grad = torch.randn_like(p) * 0.1
    opt.step(max_grad_norm=None)
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="stale block-diagonal"):
        with torch.no_grad(), graphs.capture(g2):
            net._audio_pair.encode(xa, xb)


def test_passive_training_graph_is_captured_next_to_a_live_ppo_policy():
    """PassiveTrainer captures its training step right after bumping the parameter epoch; the refresh hooks of every live
    FusedAudioPair -- a PPO policy on the device, one left on the host -- see a changed key during that capture and must stay quiet."""
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
    from m2h.rl.ppo.policy import Move2HearPolicy
    dev = _dev()
    on_device = _policy(2, dev)
    on_host = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    _fused(on_device.pol_net, *_inputs(3, 1, dev))
    tr = PassiveTrainer(passive_config(BATCH_SIZE=4, TM=32, SEED=1), dev)
    tr.setup()
    batch = tr.feeders["train"].batch()
    losses = [tr.train_batch(*batch) for _ in range(4)]      # the third call captures the step's graph, the fourth replays it
    assert all(np.isfinite(float(x)) for step in losses for x in step)
    assert on_host.pol_net._audio_pair.w is None             # nothing was built for a policy that never ran on the device
    fa, fb = _fused(on_device.pol_net, *_inputs(3, 1, dev))
    sa, sb = _separate(on_device.pol_net, *_inputs(3, 1, dev))
    assert _close(fa, sa) and _close(fb, sb)
