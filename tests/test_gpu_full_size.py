"""GPU: what used to run only inside bench.py -- the RL loop in bf16x3 arithmetic against the reference
fixtures, the headline batch (256 x 512x256) through the HIP-graph separator pair in both arithmetic modes, the full-size DD-PPO
cycle (14 envs x T=20 x 6 updates) graphs vs kernel-by-kernel, and two host threads computing in different arithmetic modes."""
import os
import threading

import numpy as np
import pytest
import torch

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach())


# ------------------------------------------------------------------------------------------------ (a) bf16x3 RL arithmetic
def test_rl_forward_in_bf16x3_math_matches_reference_fixture(golden_dir):
    """BASELINE config 5's arithmetic (every forward GEMM as bf16x3 split products) against the fp32 reference fixture: 1e-4
    rel-L1 on every feature tensor (contract 1e-3), deterministic actions unchanged."""
    from m2h import ops
    from test_gpu_rl import _obs, _policy
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "rl_forward.npz"))
    pol, _sd = _policy(int(g["seed_w"]), dev)
    N = int(g["N"])
    obs = _obs(N, int(g["seed_x"]), dev)
    masks, h0, prev = (torch.from_numpy(g[k]).to(dev) for k in ("masks", "h0", "prev_mem"))
    T = 1e-4
    with ops.math_scope(ops.MATH_BF16X3), torch.no_grad():
        pm = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(pm, mixed_audio=obs["mixed_bin_audio_mag"])
        mem = pol.get_monoFromMem_masked(mono, prev, masks)
        assert _rel(pm, g["pred_binSepMasks"]) < T and _rel(mono, g["pred_mono"]) < T and _rel(mem, g["pred_monoFromMem"]) < T
        assert _rel(pol.pol_net.visual_encoder(obs), g["visual_feats"]) < T
        assert _rel(pol.pol_net.bin_encoder(obs, pred_binSepMasks=pm), g["bin_feats"]) < T
        feats, h1 = pol.pol_net(obs, h0, masks, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem)
        assert _rel(feats, g["gru_out"]) < T and _rel(h1, g["h1"]) < T
        v2, a2, lp2, _, probs = pol.act(obs, h0, masks, deterministic=True, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem)
        assert _rel(v2, g["act_value"]) < T and _rel(probs, g["act_probs"]) < T and _rel(lp2, g["det_logp"]) < T
        assert torch.equal(a2.cpu(), torch.from_numpy(g["det_action"]))
    assert ops.math_mode() == ops.MATH_FP32


def test_updates_in_bf16x3_math_match_reference_fixture(golden_dir):
    """One PPO.update_pol and one PPO.update_sep (2 epochs each) with forward and input-gradient GEMMs in bf16x3 math: losses
    within 1e-4 of the reference's, post-step weights as close as the fp32 run's (Adam's first steps are +-lr per element)."""
    from m2h import ops
    from m2h.common.rollout_storage import RolloutStoragePol, RolloutStorageSep
    from m2h.common.spaces import move2hear_observation_space
    from test_gpu_train import _agent, _fill_pol_storage
    dev = _dev()
    gold = np.load(os.path.join(golden_dir, "rl_updates.npz"))
    with ops.math_scope(ops.MATH_BF16X3):
        agent, pol, sd = _agent(int(gold["seed_w"]), dev)
        T, N = int(gold["pol_T"]), int(gold["pol_N"])
        obs_all = {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations((T + 1) * N, int(gold["pol_obs_seed"])).items()}
        ro = RolloutStoragePol(T, N, move2hear_observation_space(), 512)
        _fill_pol_storage(ro, obs_all, T, N, torch.Generator().manual_seed(int(gold["pol_fill_seed"])))
        ro.to(dev)
        torch.manual_seed(int(gold["pol_perm_seed"]))
        v, a, h = agent.update_pol(ro)
        ref = gold["pol_losses"]
        assert abs(v - ref[0]) < 1e-4 * max(1, abs(ref[0])) and abs(a - ref[1]) < 1e-4 and abs(h - ref[2]) < 1e-4
        post = pol.state_dict()
        for key in gold.files:
            if key.startswith("polpost."):
                k = key[len("polpost."):]
                bad = (((torch.from_numpy(gold[key]) - sd[k]) - (post[k].cpu() - sd[k])).abs() > 2e-5).float().mean().item()
                assert bad < 0.02, (k, bad)
        agent2, pol2, sd2 = _agent(int(gold["seed_w"]), dev)
        T, N = int(gold["sep_T"]), int(gold["sep_N"])
        obs_s = {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations((T + 1) * N, int(gold["sep_obs_seed"])).items()}
        rs = RolloutStorageSep(T, N, move2hear_observation_space())
        g2 = torch.Generator().manual_seed(int(gold["sep_fill_seed"]))
        for k in rs.observations:
            rs.observations[k].copy_(obs_s[k].reshape(T + 1, N, *obs_s[k].shape[1:]))
        rs.prev_pred_monoFromMem.copy_(torch.rand(T + 1, N, 512, 32, 1, generator=g2))
        rs.masks.copy_((torch.rand(T + 1, N, 1, generator=g2) > 0.3).float())
        rs.to(dev)
        torch.manual_seed(int(gold["sep_perm_seed"]))
        b, m, mm = agent2.update_sep(rs)
        ref = gold["sep_losses"]
        assert abs(b - ref[0]) < 1e-4 * ref[0] and abs(m - ref[1]) < 1e-4 * ref[1] and abs(mm - ref[2]) < 1e-4 * ref[2]
        post = pol2.state_dict()
        for k in ("acoustic_mem.cnn.0.weight", "acoustic_mem.cnn.2.weight"):
            bad = (((torch.from_numpy(gold["seppost." + k]) - sd2[k]) - (post[k].cpu() - sd2[k])).abs() > 1e-4).float().mean().item()
            assert bad < 0.02, (k, bad)


# ------------------------------------------------------------------------------------------------ (b) the headline batch
@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16x3", 1e-4)])
def test_headline_batch_through_the_graphed_pair(mode, tol):
    """B = 256 x 512x256 (BASELINE config 2) through GraphedSeparatorPair, as bench.py runs it: 16 samples spread over the batch
    against the live oracle; the whole batch for neighbour-independence (rolling the batch rolls the outputs, bit for bit) and
    replay determinism."""
    from m2h import ops
    from m2h.graphs import GraphedSeparatorPair
    from test_gpu_unet import _policy
    dev = _dev()
    B, Tm = 256, 256
    pol, sd = _policy(1, dev)
    g = torch.Generator(device=dev).manual_seed(1000)
    re, im = (torch.randn(B, 512, Tm, 2, device=dev, generator=g) for _ in range(2))
    gain = torch.exp(torch.rand(B, 512, 1, 1, device=dev, generator=g) * 3.0 - 2.0)
    mix = torch.log1p(torch.sqrt(re * re + im * im) * gain).contiguous()
    del re, im
    tc = torch.randint(0, 11, (B, 1), device=dev, generator=g)
    obs = {"mixed_bin_audio_mag": mix, "target_class": tc}
    with ops.math_scope(ops.MATH_BF16X3 if mode == "bf16x3" else ops.MATH_FP32):
        pair = GraphedSeparatorPair(pol, obs)
        masks, mono = (t.clone() for t in pair())
        m2, mo2 = pair()
        assert torch.equal(masks, m2) and torch.equal(mono, mo2)                       # replay determinism
        idx = torch.arange(0, B, 16, device=dev)
        with torch.no_grad():
            o_masks, o_mono = O.passive_pair(sd, mix[idx].cpu(), tc[idx].cpu())
        torch.set_num_threads(8)
        assert _rel(masks[idx], o_masks) < tol and _rel(mono[idx], o_mono) < tol
        em = torch.expm1(mix[idx]).cpu()
        assert _rel(masks[idx].cpu() * em, o_masks * em) < tol                          # the contract metric (pred_bin)
        mix.copy_(torch.roll(mix, 37, 0))                                                # same addresses: the graph is fed by copy
        tc.copy_(torch.roll(tc, 37, 0))
        m3, mo3 = pair()
        assert torch.equal(torch.roll(masks, 37, 0), m3) and torch.equal(torch.roll(mono, 37, 0), mo3)


# ------------------------------------------------------------------------------------------------ (c) the full-size cycle
def test_full_size_ddppo_cycle_graphs_equal_kernel_by_kernel():
    """nearTarget.yaml's schedule at full size (14 envs, T = 20, 6 x (rollout + update_pol) + 6 x update_sep, 4 epochs): the
    HIP-graph run and the kernel-by-kernel run leave bit-identical storages, statistics and weights after two cycles."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    dev = _dev()
    runs = []
    for graphs in (False, True):
        tr = PPOTrainer(near_target_config(use_hip_graphs=graphs), dev)
        tr.setup()
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
        out = []
        for c in range(2):
            torch.manual_seed(300 + c)
            out.append(tr.train_cycle(log_stats=True))
        assert out[0]["env_steps"] == 6 * 20 * 14
        runs.append(([(o["pol_losses"], o["sep_losses"]) for o in out], tr.rollouts_pol.rewards.cpu().clone(), tr.rollouts_pol.actions.cpu().clone(),
                     tr.rollouts_sep.prev_pred_monoFromMem[::17].cpu().clone(), tr.stats.episode_rewards.cpu().clone(), tr.scalars,
                     {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}))
        if graphs:
            assert len(tr._graph_state.graphs) == 3 and tr.agent._pol_graph.graph is not None
    a, b = runs
    assert a[0] == b[0] and a[5] == b[5]
    for x, y in zip(a[1:5], b[1:5]):
        assert torch.equal(x, y)
    for k in a[6]:
        assert torch.equal(a[6][k], b[6][k]), k
    assert float(a[4].abs().sum()) > 0 and len(a[5]) == 12


# ------------------------------------------------------------------------------------------------ (d) no process-global arithmetic
def test_two_host_threads_compute_in_different_math_modes_side_by_side():
    """the library holds no global mutable state.  Two host threads -- one in fp32-MFMA arithmetic, one in bf16x3 --
    run separator pairs concurrently on their own streams; each must reproduce its single-thread result bit for bit."""
    from m2h import ops
    from test_gpu_unet import _policy
    dev = _dev()
    pol, _ = _policy(3, dev)
    mixed, tc = synthetic.make_passive_inputs(6, 32, 9)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

    def pair():
        with torch.no_grad():
            m = pol.get_binSepMasks(obs)
            return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
    want = {}
    for mode in (ops.MATH_FP32, ops.MATH_BF16X3):
        with ops.math_scope(mode):
            want[mode] = tuple(t.clone() for t in pair())
    assert not torch.equal(want[ops.MATH_FP32][0], want[ops.MATH_BF16X3][0])   # the two arithmetics do differ in the last bits
    torch.cuda.synchronize()
    start = threading.Barrier(2)
    errors = []

    def worker(mode):
        try:
            torch.cuda.set_device(dev)
            ops.set_math_mode(mode)
            assert _lib_mode() == mode
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                start.wait()
                for _ in range(25):
                    m, mono = pair()
                    if not (torch.equal(m, want[mode][0]) and torch.equal(mono, want[mode][1])):
                        errors.append("mode %d: result differs from the single-thread run" % mode)
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def _lib_mode():
        from m2h import _lib
        return _lib.load().m2h_get_math_mode()
    ts = [threading.Thread(target=worker, args=(m,)) for m in (ops.MATH_FP32, ops.MATH_BF16X3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert ops.math_mode() == ops.MATH_FP32 and _lib_mode() == 0   # the main thread's mode was never touched
