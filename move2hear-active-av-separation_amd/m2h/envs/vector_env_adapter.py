"""Real-environment adapter (row N4 of SURVEY 8f): the reference's ``VectorEnvCustom`` protocol on one side, the device-side
env interface of ``m2h.rl.ppo.ppo_trainer.PPOTrainer`` on the other.

The reference steps its simulator processes with ``envs.step([a[0].item() for a in actions])`` and gets back one
``(observation dict, reward, done, info)`` tuple per env, which it batches with ``batch_obs`` and turns into reward / not-done
tensors (audio_separation/rl/ppo/ppo_trainer.py:323-345, common/env_utils.py:71-528: ``num_envs``, ``observation_spaces``,
``action_spaces``, ``reset``, ``step``, ``close``).  ``HostVectorEnvAdapter`` does exactly that and hands the trainer what
``SyntheticVecEnv`` hands it: a dict of float32 device tensors, rewards ``[N,1]``, not-done masks ``[N,1]`` and the two distance
infos as ``[N,1]`` tensors.  It costs what the real simulator costs -- one device->host read of the actions and one host->device
copy of the observations per step -- and, being host-side, keeps the rollout step out of the HIP-graph path (the trainer
enqueues such a step kernel by kernel).
"""
import torch

from ..common.utils import batch_obs

INFO_KEYS = ("normalized_geo_distance_to_target_audio_source", "geo_distance_to_target_audio_source")


class HostVectorEnvAdapter:
    def __init__(self, envs, device):
        self.envs = envs
        self.device = device
        self.num_envs = envs.num_envs
        self.observation_spaces = envs.observation_spaces
        self.action_spaces = envs.action_spaces
        self.last_dones = [False] * self.num_envs   # host copy of the last step's `dones` (the trainer's episode-step counter)

    def reset(self):
        return dict(batch_obs(self.envs.reset(), self.device))

    def step(self, actions):
        """actions: [N,1] int64 device tensor (Policy.act).  -> (obs dict, rewards [N,1], not-done masks [N,1], infos)."""
        outputs = self.envs.step([int(a) for a in actions.reshape(-1).tolist()])          # ppo_trainer.py:323
        observations, rewards, dones, infos = [list(x) for x in zip(*outputs)]             # :326
        self.last_dones = [bool(d) for d in dones]
        batch = dict(batch_obs(observations, self.device))                                  # :328
        masks = torch.tensor([[0.0] if done else [1.0] for done in dones], dtype=torch.float32, device=self.device)   # :329-331
        rew = torch.tensor(rewards, dtype=torch.float32, device=self.device).reshape(-1, 1)
        info_t = {k: torch.tensor([[float(info.get(k, 0.0))] for info in infos], dtype=torch.float32, device=self.device)
                  for k in INFO_KEYS}                                                        # :332-337
        return batch, rew, masks, info_t

    def close(self):
        self.envs.close()
