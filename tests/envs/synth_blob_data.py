"""Test fixture: a contextual-bandit-like IMAGE environment with the gym-0.23 interface (module name follows the
reference's `<env_id>_data` + `env_creator` rule). The observation is a (3,96,96) fp32 image with a bright square
whose horizontal position encodes a target in [-1,1]; reward = -(a0 - target)^2 - 0.01*|a|^2. A policy can only
do better than chance by reading the position from the pixels through its conv encoder."""
import numpy as np


class _Box:
    def __init__(self, low, high, shape=None):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = tuple(shape) if shape is not None else self.low.shape
        self.dtype = np.float32


class SynthBlob:
    max_episode_steps = 20

    def __init__(self, seed=0, act_dim=3):
        self.rng = np.random.default_rng(seed)
        self.action_space = _Box(-np.ones(act_dim), np.ones(act_dim))
        self.observation_space = _Box(0.0, 1.0, shape=(3, 96, 96))
        self.t, self.target = 0, 0.0

    def seed(self, s):
        self.rng = np.random.default_rng(s)

    def _obs(self):
        img = np.full((3, 96, 96), 0.1, np.float32)
        cx = int(round((self.target + 1) / 2 * 71)) + 12
        img[:, 36:60, cx - 12:cx + 12] = 0.9
        img += self.rng.random((3, 96, 96), dtype=np.float32) * 0.02
        return img

    def reset(self):
        self.t = 0
        self.target = float(self.rng.uniform(-1, 1))
        return self._obs(), {}

    def step(self, a):
        a = np.asarray(a, np.float32).reshape(-1)
        r = -float((a[0] - self.target) ** 2) - 0.01 * float((a ** 2).sum())
        self.t += 1
        self.target = float(self.rng.uniform(-1, 1))
        trunc = self.t >= self.max_episode_steps
        return self._obs(), r, False, {"TimeLimit.truncated": trunc}

    def render(self):
        pass


def env_creator(**kwargs):
    return SynthBlob(seed=kwargs.get("seed", 0) or 0, act_dim=int(kwargs.get("action_dim", 3)))
