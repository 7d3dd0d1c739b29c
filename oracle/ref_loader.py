"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference (Jingliang-Duan/DSAC-v2).

Imports `/root/reference/dsac_v2.py` (and friends) in THIS container, where the reference is
mounted read-only, by injecting stub modules for the two packages the reference imports at
module scope but that are not installed here (`gym`, `tensorboard`).
Nothing under /root/reference is copied or modified.

Used for two things only:
  * `oracle/make_golden.py` -- generate golden input/output vectors committed under tests/golden/
  * `tests/test_oracle_vs_reference.py` -- pin the restatement in `oracle/dsact_oracle.py`
    against the live reference (skipped automatically where /root/reference is absent,
    e.g. on the GPU box).

Never imported by the product (`dsac-v2_amd/`), `bench.py` or `__graft_entry__.py`.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DSACT_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "dsac_v2.py"))


def _stub(name):
    m = types.ModuleType(name)
    m.__dict__["__stub__"] = True
    sys.modules[name] = m
    return m


class _Wrapper:
    """Functional stand-in for gym.Wrapper (gym 0.23 style): forwards to `.env`."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, item):
        if item == "env":
            raise AttributeError(item)
        return getattr(self.env, item)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, a):
        return self.env.step(a)


class _TimeLimit(_Wrapper):
    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, a):
        obs, r, done, info = self.env.step(a)
        self._elapsed_steps += 1
        if self._max_episode_steps is not None and self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, r, done, info

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)


def install_stubs():
    """Install gym/tensorboard stubs (idempotent)."""
    import numpy as np

    if not hasattr(np, "float_"):  # reference utils/common_utils.py:112 uses np.float_ (NumPy<2)
        np.float_ = np.float64
    if "gym" not in sys.modules:
        gym = _stub("gym")
        gym.Wrapper = _Wrapper
        gym.Env = object
        core = _stub("gym.core")
        core.ObsType = object
        core.ActType = object
        core.Wrapper = _Wrapper
        gym.core = core
        wr = _stub("gym.wrappers")
        tl = _stub("gym.wrappers.time_limit")
        tl.TimeLimit = _TimeLimit
        wr.time_limit = tl
        wr.TimeLimit = _TimeLimit
        gym.wrappers = wr
        ut = _stub("gym.utils")
        gym.utils = ut
    if "tensorboard" not in sys.modules:
        tb = _stub("tensorboard")
        be = _stub("tensorboard.backend")
        app = _stub("tensorboard.backend.application")
        tb.backend = be
        be.application = app


def import_reference():
    """Returns the reference's `dsac_v2` module (unmodified code, run in place)."""
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    # `utils` / `training` are packages of the reference; importing `utils` appends its own dir
    # to sys.path (reference utils/__init__.py) which the rest of the reference relies on.
    importlib.import_module("utils")
    return importlib.import_module("dsac_v2")


def reference_kwargs(obs_dim, act_dim, hidden=(256, 256, 256), act_limit=0.4, **over):
    """kwargs dict exactly as example_train/dsacv2_mlp_mujoco_offserial.py:21-141 + init_args
    (utils/init_args.py:11-83) would produce it, for synthetic shapes."""
    import numpy as np

    kw = dict(
        env_id="synthetic", algorithm="DSAC_V2", enable_cuda=False, seed=0,
        reward_scale=1, action_type="continu",
        value_func_name="ActionValueDistri", value_func_type="MLP",
        value_hidden_sizes=list(hidden), value_hidden_activation="gelu",
        value_output_activation="linear", value_min_log_std=-8, value_max_log_std=8,
        policy_func_name="StochaPolicy", policy_func_type="MLP",
        policy_act_distribution="TanhGaussDistribution",
        policy_hidden_sizes=list(hidden), policy_hidden_activation="gelu",
        policy_output_activation="linear", policy_min_log_std=-20, policy_max_log_std=0.5,
        value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4,
        gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2,
        TD_bound=1, bound=True, trainer="off_serial_trainer",
        buffer_name="replay_buffer", buffer_warm_size=1000, buffer_max_size=10000,
        replay_batch_size=256, sample_interval=1, sample_batch_size=20, noise_params=None,
        use_gpu=False, batch_size_per_sampler=20,
        obsv_dim=obs_dim, action_dim=act_dim,
        action_high_limit=np.full((act_dim,), act_limit, dtype=np.float32),
        action_low_limit=np.full((act_dim,), -act_limit, dtype=np.float32),
        additional_info={}, cnn_shared=False,
    )
    kw.update(over)
    return kw
