"""HipOffSampler -- env-stepping actor with the reference's OffSampler semantics
(reference training/off_sampler.py:12-101), acting with the LIVE learner weights.

The reference moves the whole module to the CPU around every sampling call
(ModuleOnDevice, utils/common_utils.py:164-177; training/trainer.py:63-66). Here the parameters never
leave HBM: `networks.policy(obs)` of an attached ApproxContainer runs the fused-MLP HIP forward
(dsact_policy_forward) on the batch-1 observation and returns the logits on the host, where the
tanh-Gaussian draw consumes the torch global generator exactly like TanhGaussDistribution.sample().

Environment protocol (gym 0.23 style, what the reference's wrappers provide): `reset() -> (obs, info)`
or `obs`; `step(a) -> (obs, r, done, info)` with `info["TimeLimit.truncated"]` on time-outs;
`action_space.low/high`.
"""
import time

import numpy as np
import torch

__all__ = ["HipOffSampler", "create_sampler"]

SAMPLER_TIME_KEY = "Time/Sampler time [ms]-RL iter"  # reference utils/tensorboard_setup.py:150


def _container(**kwargs):
    """`__import__(algorithm.lower()).ApproxContainer(**kwargs)` -- the reference's rule (off_sampler.py:19-23,
    evaluator.py:16-20): a plain CPU torch module until the learner's attached container replaces it"""
    module = __import__(kwargs["algorithm"].lower())
    return getattr(module, "ApproxContainer")(**kwargs)


def _reset(env):
    out = env.reset()
    if isinstance(out, tuple) and len(out) == 2 and isinstance(out[1], dict):
        return out
    return out, {}


class SampleBatch(list):
    """The reference's list of (obs, info, act, rew, next_obs, done, logp, next_info) tuples (off_sampler.py:74-84) that
    ALSO carries the same transitions as packed float32 arrays: HipReplayBuffer.add_batch hands `packed` to the HBM ring
    without walking the tuples; any other consumer sees an ordinary list."""

    packed = None   # (obs[n,O], act[n,A], rew[n], obs2[n,O], done[n], logp[n])


class HipOffSampler:
    def __init__(self, index=0, **kwargs):
        from plugin import create_env

        self.env = kwargs.get("env") or create_env(**kwargs)
        if kwargs.get("seed") is not None and hasattr(self.env, "seed"):
            self.env.seed(kwargs["seed"])  # reference set_seed(..., env) seeds with the plain seed
        self.obs, self.info = _reset(self.env)
        # The reference builds its own ApproxContainer here (off_sampler.py:19-23) -- the trainer replaces it with the
        # learner's (trainer.py:24-26), but its random initialisation consumes the torch global generator, so a run
        # from the same seed only follows the reference's trajectory if this one is built too.
        self.networks = kwargs.get("networks")
        if self.networks is None and "algorithm" in kwargs:
            self.networks = _container(**kwargs)
        self.noise_params = kwargs.get("noise_params")
        if self.noise_params is not None:
            raise NotImplementedError("exploration noise is not part of the DSAC-T path (default None)")
        self.sample_batch_size = kwargs["batch_size_per_sampler"] if "batch_size_per_sampler" in kwargs \
            else kwargs["sample_batch_size"]
        self.action_type = kwargs.get("action_type", "continu")
        self.reward_scale = kwargs.get("reward_scale", 1)
        self.total_sample_number = 0
        # additive: `hip_sampler_general_path=True` forces the general loop below (reproducibility comparisons against the
        # reference sampler: the fast path's logits differ from the module forward's in the last bits)
        self.general_path = bool(kwargs.get("hip_sampler_general_path", False))
        self._fast_ok, self._fast_started = {}, False

    def load_state_dict(self, state_dict):
        self.networks.load_state_dict(state_dict)

    def _fast_engine(self):
        """the engine behind an ATTACHED MLP policy whose action distribution is the tanh-Gaussian (what every shipped
        DSAC example uses): one dsact_act_sample call per environment step replaces tensor round trips, Normal objects and
        .cpu().numpy() (145 -> ~50 us per step, bench.py `e2e`). None: the general path below."""
        if self.general_path:
            return None
        pol = getattr(self.networks, "policy", None)
        eng = getattr(pol, "_engine", None)
        if eng is None or self.action_type != "continu" or getattr(eng, "conv_type", None):
            return None
        if type(pol).__name__ != "HipStochaPolicy":
            return None
        # the library's own gate (act_fast_ok, csrc/dsact_api.hip): MLP policy, observation <= 768 floats, at most 4 hidden
        # layers, act_dim <= 32, DSACT_NO_FAST_ACT unset -- asked once per engine, never restated here
        ok = self._fast_ok.get(id(eng))
        if ok is None:
            try:
                ok = eng.debug_get("act_fast") == 1.0
            except Exception:
                ok = False
            self._fast_ok[id(eng)] = ok
        return eng if ok else None

    def _sample_fast(self, eng):
        """same step order and the same generator consumption as the loop in sample() (ONE torch.randn(1, A) per step:
        Normal.sample() is mean + std * that draw); actions / log-probs equal the general path's within fp32 rounding
        (k_act_mlp sums a layer in another order than the module forward: 2e-6 of the action limit, 5e-4 on logp), so a
        long run from the same seed may leave the general path's trajectory. `hip_sampler_general_path=True` forces the
        general loop."""
        n, env = self.sample_batch_size, self.env
        O, A = eng.obs_dim, eng.act_dim
        obs_b, obs2_b = np.empty((n, O), np.float32), np.empty((n, O), np.float32)
        act_b = np.empty((n, A), np.float32)
        clip_b = np.empty((n, A), np.float32)
        rew_b, done_b, logp_b = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
        low, high = env.action_space.low, env.action_space.high
        batch = SampleBatch()
        append, step, scale = batch.append, env.step, self.reward_scale
        obs, info = self.obs, self.info
        self._fast_started = False
        eng.note_torch_writes(self.networks.policy.parameters())   # (host-side acting: weights written with torch ops since the last call)
        # per-step host work is what this loop costs once the acting forward runs on the host (csrc/dsact_host_act.h: ~2-9 us per
        # call): row views and their addresses are made once per call, the N(0,1) draw lands in ONE reused tensor
        # (torch.randn(1, A, out=...) consumes the generator exactly as torch.randn(1, A) does), the action is clipped with two
        # ufunc calls into a preallocated row (np.clip's wrapper costs more than the clip)
        eps_t = torch.empty(1, A)
        randn = torch.randn
        # plain integer addresses: the binding used here takes void* arguments, so no ctypes pointer object is made per step
        eps_a, obs_a, act_a, logp_a = eps_t.data_ptr(), obs_b.ctypes.data, act_b.ctypes.data, logp_b.ctypes.data
        obs_s, act_s = 4 * O, 4 * A
        act_into = eng.act_sample_addr
        flat = np.ndim(obs) == 1
        maximum, minimum = np.maximum, np.minimum
        for i in range(n):
            ob = obs_b[i]
            ob[:] = obs if flat else np.reshape(obs, -1)
            randn(1, A, out=eps_t)
            act_into(obs_a + i * obs_s, eps_a, act_a + i * act_s, logp_a + 4 * i)
            self._fast_started = True     # (an environment step follows: no silent fallback from here on)
            a_i, c_i = act_b[i], clip_b[i]
            minimum(maximum(a_i, low, out=c_i), high, out=c_i)
            next_obs, reward, done, next_info = step(c_i)
            truncated = bool(next_info.get("TimeLimit.truncated", False))
            next_info["TimeLimit.truncated"] = truncated
            if truncated:
                done = False  # time-outs are stored as non-terminal (off_sampler.py:70-73)
            ob2 = obs2_b[i]
            ob2[:] = next_obs if flat else np.reshape(next_obs, -1)
            r = scale * reward
            rew_b[i] = r
            done_b[i] = done
            append((ob if flat else ob.reshape(np.shape(obs)), info, a_i, r, ob2 if flat else ob2.reshape(np.shape(next_obs)), done,
                    logp_b[i], next_info))
            obs, info = next_obs, next_info
            if done or truncated:
                obs, info = _reset(env)
        self.obs, self.info = obs, info
        batch.packed = (obs_b, act_b, rew_b, obs2_b, done_b, logp_b)
        return batch

    def sample(self):
        self.total_sample_number += self.sample_batch_size
        t0 = time.perf_counter()
        eng = self._fast_engine()
        if eng is not None:
            from dsact._ffi import DsactError
            try:
                batch = self._sample_fast(eng)
                return batch, {SAMPLER_TIME_KEY: (time.perf_counter() - t0) * 1000}
            except DsactError as ex:
                # dsact_act_sample refused this handle (DSACT_E_INVALID) before anything was stepped: the general loop
                # serves every policy (a safety net behind the "act_fast" gate: one torch.randn draw has been spent by then).
                # Anything else (a failed launch, a hand-over timeout) is the caller's to see.
                if "E_INVALID" not in str(ex) or self._fast_started:
                    raise
                self._fast_ok[id(eng)] = False
        batch = []
        for _ in range(self.sample_batch_size):
            obs_t = torch.from_numpy(np.expand_dims(self.obs, axis=0).astype("float32"))
            with torch.no_grad():
                logits = self.networks.policy(obs_t)
                dist = self.networks.create_action_distributions(logits)
                action, logp = dist.sample()
            action = action.detach()[0].cpu().numpy()
            logp = logp.detach()[0].cpu().numpy()
            action = np.array(action)
            clipped = action.clip(self.env.action_space.low, self.env.action_space.high)
            next_obs, reward, done, next_info = self.env.step(clipped)
            truncated = bool(next_info.get("TimeLimit.truncated", False))
            next_info["TimeLimit.truncated"] = truncated
            if truncated:
                done = False  # time-outs are stored as non-terminal (off_sampler.py:70-73)
            batch.append((self.obs.copy(), self.info, action, self.reward_scale * reward, next_obs.copy(), done,
                          logp, next_info))
            self.obs, self.info = next_obs, next_info
            if done or truncated:
                self.obs, self.info = _reset(self.env)
        return batch, {SAMPLER_TIME_KEY: (time.perf_counter() - t0) * 1000}

    def get_total_sample_number(self):
        return self.total_sample_number


def create_sampler(**kwargs):
    return HipOffSampler(**kwargs)
