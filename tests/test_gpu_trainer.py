"""GPU: the DD-PPO training cycle end to end on the synthetic env (small schedule), and result-preservation of the
trainer-level re-use of separator outputs."""
import numpy as np
import pytest
import torch

from m2h import synthetic

pytestmark = pytest.mark.gpu


def _trainer(**over):
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    cfg = near_target_config(**dict(dict(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=4, use_ddppo=True), **over))
    tr = PPOTrainer(cfg, torch.device("cuda", 0))
    tr.setup()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 5).items()}
    tr.actor_critic.load_state_dict(sd)
    return tr, sd


def test_training_cycle_runs_and_updates_only_trainable_parts():
    tr, sd = _trainer()
    res = tr.train_cycle()
    assert res["env_steps"] == 2 * 4 * 3
    assert all(np.isfinite(res["pol_losses"])) and all(np.isfinite(res["sep_losses"]))
    post = tr.actor_critic.state_dict()
    changed = lambda k: not torch.equal(post[k].cpu(), sd[k])  # noqa: E731
    assert changed("pol_net.visual_encoder.cnn.0.weight") and changed("pol_net.state_encoder.rnn.weight_hh_l0")
    assert changed("action_dist.linear.weight") and changed("critic.fc.bias") and changed("acoustic_mem.cnn.0.weight")
    for k in post:
        if "Sep_" in k or "bin2mono_" in k:
            assert torch.equal(post[k].cpu(), sd[k]), k  # frozen separators untouched
    t = tr.all_reduce_stats()
    assert t.shape == (6,) and float(t[1]) == 2 * 3  # two episodes of 4 steps finished per env
    res2 = tr.train_cycle()
    assert all(np.isfinite(res2["pol_losses"]))


def test_next_step_cache_preserves_rollout_contents():
    """Re-using the next-observation separator outputs as the following step's current outputs stores exactly what a
    from-scratch recomputation stores (frozen eval-mode networks are deterministic per observation)."""
    tr_a, _ = _trainer()
    tr_b, _ = _trainer()
    for i in range(3):
        torch.manual_seed(100 + i)  # action sampling draws from the global device generator: align the two trainers
        tr_a._collect_rollout_step()
        tr_b._next_cache = None  # force recomputation every step
        torch.manual_seed(100 + i)
        tr_b._collect_rollout_step()
    for name in ("pred_binSepMasks", "pred_mono", "prev_pred_monoFromMem", "rewards", "value_preds", "actions"):
        assert torch.equal(getattr(tr_a.rollouts_pol, name), getattr(tr_b.rollouts_pol, name)), name
