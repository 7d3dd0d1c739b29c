"""Benchmark / test fixture: an environment with gym_humanoid's SHAPES (obs 376, act 17, limits +-0.4,
env_gym/gym_humanoid_data.py:6) and the gym-0.23 interface the reference's samplers expect; module name follows the
reference's `<env_id>_data` + `env_creator` rule. The dynamics are a table lookup (a pre-generated pool of
observations, reward = -|a|^2, 1000-step time limit): an env step costs ~2 us, so an end-to-end measurement around it
shows the framework's own cost per environment step, not MuJoCo's."""
import numpy as np


class _Box:
    def __init__(self, low, high):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32


class SynthHumanoid:
    max_episode_steps = 1000
    O, A = 376, 17

    def __init__(self, seed=0):
        self.action_space = _Box(np.full(self.A, -0.4), np.full(self.A, 0.4))
        self.observation_space = _Box(np.full(self.O, -np.inf), np.full(self.O, np.inf))
        self.seed(seed)

    def seed(self, s):
        rng = np.random.default_rng(s)
        self.pool = rng.standard_normal((4096, self.O), dtype=np.float32)
        self.i, self.t = int(rng.integers(0, 4096)), 0

    def reset(self):
        self.t = 0
        self.i = (self.i + 17) & 4095
        return self.pool[self.i], {}

    def step(self, a):
        a = np.asarray(a, np.float32).reshape(-1)
        self.i = (self.i + 1) & 4095
        self.t += 1
        return self.pool[self.i], -float(a @ a), False, {"TimeLimit.truncated": self.t >= self.max_episode_steps}

    def render(self):
        pass


def env_creator(**kwargs):
    return SynthHumanoid(seed=kwargs.get("seed", 0) or 0)
