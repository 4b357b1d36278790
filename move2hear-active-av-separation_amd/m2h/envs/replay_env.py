"""Table-driven replay environment: a vector env whose whole trajectory is a pure function of (seed, actions).

Stands where the reference's ``VectorEnvCustom`` stands (audio_separation/common/env_utils.py:71-528) when the SAME episode
has to be run by two different programs -- the reference's own training loop on the CPU (oracle/gen_trainer_golden.py) and the
m2h trainer on the GPU -- so that their stored rollouts, rewards and statistics can be compared value by value.  Unlike
``SyntheticVecEnv`` (pools drawn from the device generator) everything here comes from numpy's PCG64:

  * a pool of ``pool`` observations (``m2h.synthetic.make_rl_observations``: the reference's sensors and shapes,
    config/default.py:130-157);
  * env ``e`` is in state ``s[e]`` (a pool index); a step with action ``a`` moves it to ``(5 s + a + 1 + e) mod pool`` -- the
    trajectory depends on the actions taken, as a simulator's does;
  * episode ``j`` of env ``e`` lasts ``episode_lens[j, e]`` steps (all ``episode_len`` in lockstep mode, the assumption of
    ppo_trainer.py:383-384; drawn from [2, episode_len] with ``ragged``) and is followed by an auto-reset
    (env_utils.py:186-187) into ``reset_states[j + 1, e]``;
  * the env's own reward (used when RL.PPO.nav_reward_weight = 1, farTarget.yaml) and the two distance infos read by
    ppo_trainer.py:332-337 are small rational functions of (state, action, env), exact in fp32 on either side.

``ReplayHostVecEnv`` speaks the reference's host protocol (``reset()`` -> list of observation dicts, ``step(list of int)`` ->
list of ``(observation, reward, done, info)``); ``ReplayVecEnv`` is the same world resident on the GPU behind the interface of
``SyntheticVecEnv`` (``step_device``: capturable in a HIP graph; lockstep episodes only).
"""
import numpy as np

from .. import synthetic
from ..common.spaces import Discrete, move2hear_observation_space

INFO_KEYS = ("normalized_geo_distance_to_target_audio_source", "geo_distance_to_target_audio_source")
SENSORS = ("rgb", "depth", "mixed_bin_audio_mag", "gt_bin_comps", "gt_mono_comps", "target_class")


class ReplayTables:
    """The seeded tables both env classes run from."""

    def __init__(self, num_envs, seed=0, episode_len=20, pool=16, ragged=False, max_episodes=64):
        self.num_envs, self.pool, self.episode_len = num_envs, pool, episode_len
        self.obs = synthetic.make_rl_observations(pool, 1000 + int(seed))
        r = np.random.Generator(np.random.PCG64([int(seed), 0x5EED]))
        self.reset_states = r.integers(0, pool, size=(max_episodes, num_envs)).astype(np.int64)
        if ragged:
            self.episode_lens = r.integers(2, episode_len + 1, size=(max_episodes, num_envs)).astype(np.int64)
        else:
            self.episode_lens = np.full((max_episodes, num_envs), episode_len, np.int64)
        self.ragged = ragged


def next_state(s, a, e, pool):
    return (5 * s + a + 1 + e) % pool


def env_reward(s, a, e):
    """fp32: ((7 s + 3 a + e) mod 11) / 11 - 0.5 (evaluated on the state BEFORE the step)."""
    return np.float32((7 * s + 3 * a + e) % 11) / np.float32(11.0) - np.float32(0.5)


def env_infos(s):
    """fp32 distances of the state AFTER the step: normalised (s mod 7) / 7, absolute (s mod 5)."""
    return np.float32(s % 7) / np.float32(7.0), np.float32(s % 5)


class ReplayHostVecEnv:
    def __init__(self, num_envs, seed=0, episode_len=20, pool=16, ragged=False, env_rewards=False, max_episodes=64):
        self.tab = ReplayTables(num_envs, seed, episode_len, pool, ragged, max_episodes)
        self.num_envs = num_envs
        self.env_rewards = env_rewards
        self.observation_spaces = [move2hear_observation_space()] * num_envs
        self.action_spaces = [Discrete(3)] * num_envs
        self.actions_seen = []
        self.reset()

    def _obs(self, e):
        s = int(self.s[e])
        return {k: self.tab.obs[k][s] for k in SENSORS}

    def reset(self):
        self.episode = np.zeros(self.num_envs, np.int64)
        self.t = np.zeros(self.num_envs, np.int64)
        self.s = self.tab.reset_states[0].copy()
        return [self._obs(e) for e in range(self.num_envs)]

    def step(self, actions):
        assert len(actions) == self.num_envs
        self.actions_seen.append([int(a) for a in actions])
        out = []
        for e, a in enumerate(actions):
            a, s = int(a), int(self.s[e])
            reward = float(env_reward(s, a, e)) if self.env_rewards else 0.0
            self.t[e] += 1
            done = bool(self.t[e] >= self.tab.episode_lens[self.episode[e], e])
            if done:
                self.episode[e] += 1
                self.t[e] = 0
                self.s[e] = self.tab.reset_states[self.episode[e], e]
            else:
                self.s[e] = next_state(s, a, e, self.tab.pool)
            ndg, dg = env_infos(int(self.s[e]))
            out.append((self._obs(e), reward, done, {INFO_KEYS[0]: float(ndg), INFO_KEYS[1]: float(dg)}))
        return out

    def current_episodes(self):
        """What the reference's evaluation loop reads of ``VectorEnvCustom.current_episodes()`` (ppo_trainer.py:1207-1213, :1440-1460):
        scene id, episode id, goals / start position (none here)."""
        from types import SimpleNamespace
        return [SimpleNamespace(scene_id="replay/scene%d/scene%d.glb" % (e, e), episode_id=str(int(self.episode[e])), goals=[], info=[],
                                start_position=[0.0, 0.0, 0.0]) for e in range(self.num_envs)]

    def close(self):
        pass


class ReplayVecEnv:
    """The replay world on the GPU, lockstep episodes: same attributes and methods as ``SyntheticVecEnv``."""

    def __init__(self, num_envs, device, seed=0, episode_len=20, pool=16, env_rewards=False, max_episodes=64):
        import torch
        self.tab = ReplayTables(num_envs, seed, episode_len, pool, False, max_episodes)
        self.num_envs, self.device, self.episode_len, self.env_rewards = num_envs, device, episode_len, env_rewards
        self.observation_spaces = [move2hear_observation_space()] * num_envs
        self.action_spaces = [Discrete(3)] * num_envs
        self.pools = {k: torch.from_numpy(np.ascontiguousarray(self.tab.obs[k])).float().to(device) for k in SENSORS}
        self.reset_states = torch.from_numpy(self.tab.reset_states).to(device)
        self.s = torch.zeros(num_envs, dtype=torch.int64, device=device)
        self.episode = torch.zeros(1, dtype=torch.int64, device=device)   # device-resident: a replayed graph advances it
        self.env_ids = torch.arange(num_envs, dtype=torch.int64, device=device)
        self.zero = torch.zeros(num_envs, 1, device=device)
        self.one = torch.ones(num_envs, 1, device=device)
        self._g = torch.Generator(device=device)
        self.t = 0
        self.include_phase = None

    @property
    def generator(self):
        return self._g

    def _obs(self):
        return {k: p.index_select(0, self.s) for k, p in self.pools.items()}

    def reset(self):
        self.t = 0
        self.episode.zero_()
        self.s.copy_(self.reset_states[0])
        return self._obs()

    def step(self, actions):
        done = self.t + 1 >= self.episode_len
        out = self.step_device(actions, done)
        self.t = 0 if done else self.t + 1
        return out

    def step_device(self, actions, done):
        import torch
        a = actions.reshape(-1)
        if self.env_rewards:
            k = (7 * self.s + 3 * a + self.env_ids) % 11
            rewards = (k.float() / 11.0 - 0.5).view(-1, 1)
        else:
            rewards = self.zero
        if done:
            self.episode.add_(1)
            self.s.copy_(self.reset_states.index_select(0, self.episode).view(-1))
        else:
            self.s.copy_((5 * self.s + a + 1 + self.env_ids) % self.tab.pool)
        infos = {INFO_KEYS[0]: ((self.s % 7).float() / 7.0).view(-1, 1), INFO_KEYS[1]: (self.s % 5).float().view(-1, 1)}
        return self._obs(), rewards, (self.zero if done else self.one), infos

    def close(self):
        pass
