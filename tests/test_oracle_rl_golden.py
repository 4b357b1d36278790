"""CPU: the oracle's RL-path restatements against fixtures produced by the reference (gen_golden.py rl_*)."""
import os

import numpy as np
import torch

import m2h_oracle as O
from m2h import synthetic


def _sd(seed):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}


def _obs(n, seed):
    return {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations(n, seed).items()}


def _close(a, b, tol=1e-5):
    return O.rel_l1(torch.as_tensor(a), torch.as_tensor(b)) < tol


def test_rl_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "rl_forward.npz"))
    sd = _sd(int(g["seed_w"]))
    N = int(g["N"])
    obs = _obs(N, int(g["seed_x"]))
    masks, h0, prev = torch.from_numpy(g["masks"]), torch.from_numpy(g["h0"]), torch.from_numpy(g["prev_mem"])
    with torch.no_grad():
        pm, mono = O.passive_pair(sd, obs["mixed_bin_audio_mag"], obs["target_class"])
        assert _close(pm, g["pred_binSepMasks"]) and _close(mono, g["pred_mono"])
        mem = O.acoustic_mem(sd, mono, O.mask_prev_mem(prev, masks))
        assert _close(mem, g["pred_monoFromMem"])
        assert _close(O.visual_cnn(sd, obs["rgb"]), g["visual_feats"])
        assert _close(O.audio_cnn(sd, O.BIN, mixed_bin_audio_mag=obs["mixed_bin_audio_mag"], pred_binSepMasks=pm), g["bin_feats"])
        assert _close(O.audio_cnn(sd, O.MNM, pred_monoNmonoFromMem=torch.cat((mono, mem), 3)), g["mnm_feats"])
        feats, h1, _ = O.policy_net(sd, obs, h0, masks, pm, mono, mem)
        assert _close(feats, g["gru_out"]) and _close(h1, g["h1"])
        gen = torch.Generator().manual_seed(int(g["act_seed"]))
        v, a, lp, hh, probs = O.act(sd, obs, h0, masks, pm, mono, mem, generator=None if True else gen)
        # sampling consumes the global CPU generator in the reference: reproduce through it
        torch.manual_seed(int(g["act_seed"]))
        v, a, lp, hh, probs = O.act(sd, obs, h0, masks, pm, mono, mem)
        assert torch.equal(a, torch.from_numpy(g["act_action"]))
        assert _close(v, g["act_value"]) and _close(lp, g["act_logp"]) and _close(probs, g["act_probs"])
        v2, a2, lp2, _, _ = O.act(sd, obs, h0, masks, pm, mono, mem, deterministic=True)
        assert torch.equal(a2, torch.from_numpy(g["det_action"])) and _close(lp2, g["det_logp"])
        T, n = 3, 2
        obs_seq = {k: v_[:T * n] for k, v_ in obs.items()}
        ev, elp, eent, eh = O.evaluate_actions(sd, obs_seq, h0[:, :n], torch.from_numpy(g["eval_masks"]),
                                               torch.from_numpy(g["eval_actions"]), pm[:T * n], mono[:T * n], mem[:T * n])
        assert _close(ev, g["eval_value"]) and _close(elp, g["eval_logp"]) and _close(eh, g["eval_h"])
        assert abs(eent.item() - float(g["eval_entropy"])) < 1e-6


def test_sampling_contract_bit_exact(golden_dir):
    """(probs, seed) -> actions: torch.multinomial(probs, 1, True) on the CPU generator (common/utils.py:17-18)."""
    g = np.load(os.path.join(golden_dir, "rl_forward.npz"))
    torch.manual_seed(int(g["sample_seed"]))
    a = torch.multinomial(torch.from_numpy(g["sample_probs"]), 1, True)
    assert torch.equal(a, torch.from_numpy(g["sample_actions"]))
    # the draw as "argmax(probs / Exp(1) noise)" with the noise handed in (m2h_oracle_trainer.draw_actions: what the fused-sampling
    # parity test feeds with the noise the heads kernel recorded): the generator's own Exp(1) draw at the same stream position gives the
    # reference's actions, i.e. the noise is ALL the draw takes from the generator
    import m2h_oracle_trainer as OT
    torch.manual_seed(int(g["sample_seed"]))
    probs = torch.from_numpy(g["sample_probs"])
    q = torch.empty_like(probs).exponential_(1)
    assert torch.equal(OT.draw_actions(probs, q), torch.from_numpy(g["sample_actions"]))


def test_returns_advantages_generators(golden_dir):
    g = np.load(os.path.join(golden_dir, "rl_scalars.npz"))
    rewards, vp, masks, nv = (torch.from_numpy(g[k]) for k in ("rewards", "value_preds", "masks", "next_value"))
    ret, vp2 = O.compute_returns(rewards, vp, masks, nv, True, 0.99, 0.95)
    assert torch.allclose(ret, torch.from_numpy(g["returns_gae"]), atol=1e-6)
    adv = O.get_advantages(ret, vp2)
    assert torch.allclose(adv, torch.from_numpy(g["advantages"]), atol=1e-6)
    ret2, _ = O.compute_returns(rewards, vp, masks, nv, False, 0.99, 0.95)
    assert torch.allclose(ret2, torch.from_numpy(g["returns_nogae"]), atol=1e-6)
    # generator order: perm = torch.randperm(N) on the CPU generator; flattened index [t*Nsel + j] <- [t, perm[j]]
    T, N = int(g["T"]), int(g["N"])
    torch.manual_seed(int(g["gen_seed"]))
    perm = torch.randperm(N)
    expect = (torch.arange(T).reshape(T, 1) * N + perm.reshape(1, N)).reshape(T * N, 1)
    assert torch.equal(expect, torch.from_numpy(g["gen_actions_flat"]))
    torch.manual_seed(int(g["gen_sep_seed"]))
    perm = torch.randperm(5)
    expect = (torch.arange(6).reshape(6, 1) * 5 + perm.reshape(1, 5)).reshape(30, 1).float()
    assert torch.equal(expect, torch.from_numpy(g["gen_sep_masks_flat"]))
    # distributed advantages with world=1 equal the biased-variance formula
    d = O.get_advantages_distributed([ret[:-1] - vp2[:-1]])[0]
    a = ret[:-1] - vp2[:-1]
    assert torch.allclose(d, (a - a.mean()) / (a.var(unbiased=False).sqrt() + 1e-5), atol=1e-6)


def test_stft_l2_and_rewards(golden_dir):
    g = np.load(os.path.join(golden_dir, "rl_scalars.npz"))
    obs = _obs(5, int(g["l2_seed_x"]))
    d_bin, d_mono = O.stft_l2_distance(obs["mixed_bin_audio_mag"], torch.from_numpy(g["l2_masks"]), obs["gt_bin_comps"],
                                       torch.from_numpy(g["l2_mono"]), obs["gt_mono_comps"])
    assert torch.allclose(d_bin, torch.from_numpy(g["stft_l2_bin"]), rtol=1e-5)
    assert torch.allclose(d_mono, torch.from_numpy(g["stft_l2_mono"]), rtol=1e-5)
    _, gt_mono_mag = O.gt_mags(obs)
    dones = g["rew_dones"].tolist()
    nxt, cur = torch.from_numpy(g["rew_mem_next"]), torch.from_numpy(g["rew_mem_cur"])
    r1 = O.override_rewards([0.0] * 5, dones, nxt, gt_mono_mag, "quality_improvement", cur, gt_mono_mag)
    r2 = O.override_rewards([0.0] * 5, dones, nxt, gt_mono_mag)
    assert np.allclose(r1, g["rew_quality_improvement"], rtol=1e-6, atol=1e-9)
    assert np.allclose(r2, g["rew_extra"], rtol=1e-6, atol=1e-9)
