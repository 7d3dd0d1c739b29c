"""Test fixture: Pendulum-v1 dynamics with the gym-0.23 interface the reference's samplers expect
(module name follows the reference's `<env_id>_data` + `env_creator` discovery rule)."""
import numpy as np


class _Box:
    def __init__(self, low, high):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32


class SynthPendulum:
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0
    max_episode_steps = 200

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed)
        self.action_space = _Box([-self.max_torque], [self.max_torque])
        self.observation_space = _Box([-1, -1, -self.max_speed], [1, 1, self.max_speed])
        self.th, self.thdot, self.t = 0.0, 0.0, 0

    def seed(self, s):
        self.rng = np.random.default_rng(s)

    def _obs(self):
        return np.array([np.cos(self.th), np.sin(self.th), self.thdot], np.float32)

    def reset(self):
        self.th = self.rng.uniform(-np.pi, np.pi)
        self.thdot = self.rng.uniform(-1, 1)
        self.t = 0
        return self._obs(), {}

    def step(self, a):
        u = float(np.clip(np.asarray(a).reshape(-1)[0], -self.max_torque, self.max_torque))
        ang = ((self.th + np.pi) % (2 * np.pi)) - np.pi
        cost = ang ** 2 + 0.1 * self.thdot ** 2 + 0.001 * u ** 2
        self.thdot = float(np.clip(self.thdot + (3 * self.g / (2 * self.l) * np.sin(self.th)
                                                   + 3.0 / (self.m * self.l ** 2) * u) * self.dt,
                                   -self.max_speed, self.max_speed))
        self.th = self.th + self.thdot * self.dt
        self.t += 1
        trunc = self.t >= self.max_episode_steps
        return self._obs(), -cost, False, {"TimeLimit.truncated": trunc}

    def render(self):
        pass


def env_creator(**kwargs):
    env = SynthPendulum(seed=kwargs.get("seed", 0) or 0)
    if kwargs.get("max_episode_steps"):
        # the reference wraps the env in TimeLimit(max_episode_steps) (utils/wrapping_env.py:102-105); flagging the same
        # step here makes the bare env (GPU box: no reference, no wrappers) behave like the wrapped one
        env.max_episode_steps = int(kwargs["max_episode_steps"])
    return env
