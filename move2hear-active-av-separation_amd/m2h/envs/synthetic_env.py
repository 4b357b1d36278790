"""Synthetic vector environment: the Habitat/SoundSpaces loop of the reference replaced by an on-device feeder.

Stands in for ``VectorEnvCustom`` + ``HabitatSimAudioEnabledTrain`` (audio_separation/common/env_utils.py:71-528,
habitat_audio/simulator_train.py:386-486), which need the simulator and Matterport data.  What is kept from them:
  * observation dict with the reference's sensors and shapes (config/default.py:130-157);
  * cached RGB-D frames indexed by (node, angle) like ``_frame_cache`` (simulator_train.py:216-227);
  * action set MOVE_FORWARD / TURN_LEFT / TURN_RIGHT, fixed-length episodes with auto-reset (env_utils.py:186-187):
    every env is done after ``episode_len`` steps (20 nearTarget / 80 farTarget);
  * per-rank seeding ``SEED + rank * NUM_PROCESSES`` (ppo_trainer.py:609-611).
Everything lives on the GPU: step() takes the device action tensor and returns device tensors, so the rollout loop has no
host<->device synchronisation (the reference pays ``actions[i].item()`` + pipes + ``batch_obs`` H2D per step).
The audio pool holds log1p-magnitude / phase tensors with the statistics of the feeder (dataset.py:190-228); generating
them from RIR-convolved waveforms on the GPU is row N1 of SURVEY 8f.
"""
import torch

from ..common.spaces import Discrete, move2hear_observation_space


class SyntheticVecEnv:
    def __init__(self, num_envs, device, seed=0, episode_len=20, tm=32, n_freq=512, audio_pool=64, n_nodes=64):
        self.num_envs = num_envs
        self.device = device
        self.episode_len = episode_len
        self.observation_spaces = [move2hear_observation_space(tm, n_freq)] * num_envs
        self.action_spaces = [Discrete(3)] * num_envs
        g = torch.Generator(device=device).manual_seed(int(seed))
        P = audio_pool
        gain = torch.exp(torch.rand(P, n_freq, 1, 1, device=device, generator=g) * 3.0 - 2.0)
        re, im = (torch.randn(P, n_freq, tm, 2, device=device, generator=g) for _ in range(2))
        self.pool_mixed = torch.log1p(torch.sqrt(re * re + im * im) * gain).contiguous()
        gb = torch.empty(P, n_freq, tm, 8, device=device)
        gm = torch.empty(P, n_freq, tm, 4, device=device)
        for j in range(0, 8, 2):
            gb[..., j] = torch.log1p(torch.randn(P, n_freq, tm, device=device, generator=g).abs())
            gb[..., j + 1] = (torch.rand(P, n_freq, tm, device=device, generator=g) * 2 - 1) * 3.14159265
        for j in range(0, 4, 2):
            gm[..., j] = torch.log1p(torch.randn(P, n_freq, tm, device=device, generator=g).abs())
            gm[..., j + 1] = (torch.rand(P, n_freq, tm, device=device, generator=g) * 2 - 1) * 3.14159265
        self.pool_gt_bin, self.pool_gt_mono = gb, gm
        self.pool_class = torch.randint(0, 11, (P, 1), device=device, generator=g).float()
        self.n_nodes = n_nodes
        self.frames_rgb = torch.randint(0, 256, (n_nodes * 4, 128, 128, 3), device=device, generator=g).float()
        self.frames_depth = torch.rand(n_nodes * 4, 128, 128, 1, device=device, generator=g)
        self._g = g
        self.audio_idx = torch.randint(0, P, (num_envs,), device=device, generator=g)
        self.node = torch.randint(0, n_nodes, (num_envs,), device=device, generator=g)
        self.angle = torch.randint(0, 4, (num_envs,), device=device, generator=g)
        self.t = 0  # all envs step in lockstep: host-side counter, no sync
        self._consts = None
        self.pool = P
        # eval-only sensor (config/default.py: MIXED_BIN_AUDIO_PHASE_SENSOR is added by the eval configs): phases of the mixture
        self.include_phase = False
        self.pool_mixed_phase = ((torch.rand(P, n_freq, tm, 2, device=device, generator=torch.Generator(device=device).manual_seed(int(seed) + 7)) * 2 - 1)
                                 * 3.14159265).contiguous()

    def _obs(self):
        names = ["rgb", "depth", "mixed_bin_audio_mag", "gt_bin_comps", "gt_mono_comps", "target_class"]
        pools = [(self.frames_rgb, 0), (self.frames_depth, 0), (self.pool_mixed, 1), (self.pool_gt_bin, 1), (self.pool_gt_mono, 1),
                 (self.pool_class, 1)]
        if self.include_phase:
            names.insert(0, "mixed_bin_audio_phase")
            pools.insert(0, (self.pool_mixed_phase, 1))
        if self.node.is_cuda:   # one launch for the whole lookup (m2h_synth_env_observe)
            from .. import ops
            return dict(zip(names, ops.synth_env_observe(pools, self.node, self.angle, self.audio_idx)))
        f = self.node * 4 + self.angle
        return {n: p.index_select(0, f if kind == 0 else self.audio_idx) for n, (p, kind) in zip(names, pools)}

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        """actions: [N,1] int64 on the device.  Returns (obs dict, rewards [N,1], not_done masks [N,1], infos dict)."""
        done = self.t + 1 >= self.episode_len
        out = self.step_device(actions, done)
        self.t = 0 if done else self.t + 1
        return out

    def step_device(self, actions, done):
        """The device work of one step for a host-known ``done`` (all envs run fixed-length episodes in lockstep).  State
        tensors are updated in place and nothing host-side changes, so the trainer can capture this in a HIP graph
        (ppo_trainer.py); ``step`` = step_device + the host-side episode counter."""
        a = actions.reshape(-1)
        fwd = (a == 0).long() if not self.node.is_cuda else None
        if done:  # auto-reset: a new mixture and pose for every env
            self.audio_idx.copy_(torch.randint(0, self.pool, (self.num_envs,), device=self.device, generator=self._g))
            self.node.copy_(torch.randint(0, self.n_nodes, (self.num_envs,), device=self.device, generator=self._g))
            self.angle.copy_(torch.randint(0, 4, (self.num_envs,), device=self.device, generator=self._g))
        elif self.node.is_cuda:
            from .. import ops
            ops.synth_env_step(a.contiguous(), self.node, self.angle, self.n_nodes)
        else:
            self.node.copy_((self.node + fwd) % self.n_nodes)
            self.angle.copy_((self.angle + (a == 1).long() + 3 * (a == 2).long()) % 4)
        # constants (read-only for the caller): not-done flags, the zero navigation reward (weight 0 in nearTarget.yaml:48-49)
        # and the two distance infos
        c = self._consts
        if c is None:
            z = torch.zeros(self.num_envs, 1, device=self.device)
            c = self._consts = {"zero": z, "one": torch.ones(self.num_envs, 1, device=self.device),
                                "infos": {"normalized_geo_distance_to_target_audio_source": z, "geo_distance_to_target_audio_source": z}}
        return self._obs(), c["zero"], (c["zero"] if done else c["one"]), c["infos"]

    @property
    def generator(self):
        """The env's device generator (a HIP graph that contains a reset must register it)."""
        return self._g

    def close(self):
        pass


class SyntheticHostVecEnv:
    """The same synthetic world behind the reference's HOST-side vector-env protocol (common/env_utils.py:71-528): ``reset()``
    returns a list of per-env numpy observation dicts, ``step(list of int actions)`` a list of ``(observation, reward, done,
    info)`` tuples.  Stands where ``VectorEnvCustom`` stands, so the real-env adapter (vector_env_adapter.py) and the
    trainer's host-env path can be exercised without Habitat."""

    def __init__(self, num_envs, device, **kw):
        self._env = SyntheticVecEnv(num_envs, device, **kw)
        self.num_envs = num_envs
        self.observation_spaces = self._env.observation_spaces
        self.action_spaces = self._env.action_spaces

    def _split(self, batch):
        host = {k: v.cpu().numpy() for k, v in batch.items()}
        return [{k: host[k][i] for k in host} for i in range(self.num_envs)]

    def reset(self):
        return self._split(self._env.reset())

    def step(self, actions):
        dev = self._env.device
        batch, rewards, masks, infos = self._env.step(torch.tensor(actions, dtype=torch.int64, device=dev).reshape(-1, 1))
        obs, r, m = self._split(batch), rewards.reshape(-1).tolist(), masks.reshape(-1).tolist()
        inf = {k: v.reshape(-1).tolist() for k, v in infos.items()}
        return [(obs[i], r[i], m[i] == 0.0, {k: inf[k][i] for k in inf}) for i in range(self.num_envs)]

    def close(self):
        self._env.close()
