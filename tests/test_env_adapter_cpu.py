"""CPU: the real-env adapter (row N4 of SURVEY 8f) -- the reference's host-side vector-env protocol batched the way
ppo_trainer.py:323-345 batches it.  Pure host logic; the training run through it is a -m gpu test."""
import numpy as np
import torch

from m2h.envs.synthetic_env import SyntheticHostVecEnv, SyntheticVecEnv
from m2h.envs.vector_env_adapter import INFO_KEYS, HostVectorEnvAdapter


def test_adapter_batches_observations_rewards_and_done_flags():
    dev = torch.device("cpu")
    N = 3
    host = SyntheticHostVecEnv(N, dev, seed=4, episode_len=2, audio_pool=4, n_nodes=4)
    twin = SyntheticVecEnv(N, dev, seed=4, episode_len=2, audio_pool=4, n_nodes=4)   # the same world, device-side interface
    ad = HostVectorEnvAdapter(host, dev)
    assert ad.num_envs == N and len(ad.observation_spaces) == N and ad.action_spaces[0].n == 3
    obs0 = host.reset()
    assert isinstance(obs0, list) and len(obs0) == N and isinstance(obs0[0]["rgb"], np.ndarray) and obs0[0]["rgb"].shape == (128, 128, 3)
    b0, t0 = ad.reset(), twin.reset()
    assert set(b0) == set(t0)
    for k in b0:
        assert b0[k].dtype == torch.float32 and torch.equal(b0[k], t0[k].float()), k
    for step in range(3):
        actions = torch.tensor([[0], [1], [2]], dtype=torch.int64)
        batch, rew, masks, infos = ad.step(actions)
        tb, tr, tm, _ = twin.step(actions)
        for k in batch:
            assert torch.equal(batch[k], tb[k].float()), (step, k)
        assert rew.shape == (N, 1) and torch.equal(rew, tr) and torch.equal(masks, tm)
        assert float(masks.sum()) == (0.0 if step == 1 else float(N))              # episodes of 2 steps end together
        assert set(infos) == set(INFO_KEYS) and all(v.shape == (N, 1) for v in infos.values())
    ad.close()
