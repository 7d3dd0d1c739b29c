"""DSAC_V2_HIP -- drop-in algorithm plugin for the DSAC-T update on MI355X.

Discovered exactly like the reference's algorithms: `create_alg(algorithm="DSAC_V2_HIP", **kwargs)`
imports module `dsac_v2_hip` and instantiates class `DSAC_V2_HIP`; samplers / evaluators / the
PolicyRunner re-import `ApproxContainer` from the same module (reference
utils/initialization.py:48-63, training/off_sampler.py:19-23, training/evaluator.py:16-20,
utils/sys_run.py:574-578). The surface mirrors reference dsac_v2.py:19-138:

    .networks            ApproxContainer (nn.Module; state_dict keys/order == reference App. C)
    .local_update(data, iteration) -> dict with the 15 tb_info keys (dsac_v2.py:188-204)
    .get_remote_update_info(data, iteration) -> (tb_info, update_info)
    .remote_update(update_info)
    .adjustable_parameters

All arithmetic of the update runs in libdsact.so (hand-written gfx950 kernels, C-ABI in
include/dsact.h). torch owns the parameter/optimizer arenas and the checkpoint I/O only. There is no
CPU fallback for the update: constructing DSAC_V2_HIP without the library or without a GPU raises.
"""
__all__ = ["ApproxContainer", "DSAC_V2_HIP", "TanhGaussDistribution", "GaussDistribution"]

import copy
import math
import os
import sys
import time
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.abspath(__file__))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from dsact.engine import STAT_KEYS, DsactEngine, register_engine  # noqa: E402
from dsact.layout import CONV_TYPES, ArenaLayout, CnnArenaLayout  # noqa: E402

ALG_TIME_KEY = "Time/Algorithm time [ms]-RL iter"  # reference utils/tensorboard_setup.py:149
_LOG_EPS = 1e-6  # reference utils/act_distribution_cls.py:3


# --------------------------------------------------------------------------------------------------
# action distribution (host side, tiny tensors: acting / evaluation only)
# --------------------------------------------------------------------------------------------------
class TanhGaussDistribution:
    """tanh-squashed diagonal Gaussian with the interface the reference's samplers use
    (utils/act_distribution_cls.py:21-79): sample / rsample / log_prob / mode / entropy."""

    def __init__(self, logits: torch.Tensor):
        self.logits = logits
        self.mean, self.std = torch.chunk(logits, chunks=2, dim=-1)
        self.act_high_lim = torch.tensor([1.0])
        self.act_low_lim = torch.tensor([-1.0])

    def _half_range(self):
        return (self.act_high_lim - self.act_low_lim) / 2

    def _center(self):
        return (self.act_high_lim + self.act_low_lim) / 2

    def _base_log_prob(self, x):
        var = self.std ** 2
        return (-((x - self.mean) ** 2) / (2 * var) - self.std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1)

    def _squash(self, x):
        t = torch.tanh(x)
        act = self._half_range() * t + self._center()
        logp = (self._base_log_prob(x) - torch.log(1 + _LOG_EPS - t.pow(2)).sum(-1)
                - torch.log(self._half_range()).sum(-1))
        return act, logp

    def sample(self):
        with torch.no_grad():
            x = torch.normal(self.mean, self.std)  # same generator consumption as Normal.sample
        return self._squash(x)

    def rsample(self):
        eps = torch.randn(self.mean.shape, dtype=self.mean.dtype, device=self.mean.device)
        return self._squash(self.mean + eps * self.std)

    def log_prob(self, action_limited):
        x = torch.atanh((1 - _LOG_EPS) * (2 * action_limited - (self.act_high_lim + self.act_low_lim))
                        / (self.act_high_lim - self.act_low_lim))
        return self._base_log_prob(x) - torch.log(
            (self.act_high_lim - self.act_low_lim) * (1 + _LOG_EPS - torch.tanh(x).pow(2))).sum(-1)

    def entropy(self):
        return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(self.std)).sum(-1)

    def mode(self):
        return self._half_range() * torch.tanh(self.mean) + self._center()


class GaussDistribution(TanhGaussDistribution):
    """plain diagonal Gaussian (reference utils/act_distribution_cls.py:82-115; policy_act_distribution =
    "GaussDistribution"): no squashing; the action limits only bound mode()."""

    def sample(self):
        with torch.no_grad():
            x = torch.normal(self.mean, self.std)  # same generator consumption as Normal.sample
        return x, self._base_log_prob(x)

    def rsample(self):
        eps = torch.randn(self.mean.shape, dtype=self.mean.dtype, device=self.mean.device)
        x = self.mean + eps * self.std
        return x, self._base_log_prob(x)

    def log_prob(self, action):
        return self._base_log_prob(action)

    def mode(self):
        return torch.clamp(self.mean, self.act_low_lim, self.act_high_lim)


ACT_DISTRIBUTIONS = {"TanhGaussDistribution": (0, TanhGaussDistribution), "GaussDistribution": (1, GaussDistribution)}


# --------------------------------------------------------------------------------------------------
# networks: torch modules whose parameters become views of the HIP arenas once attached
# --------------------------------------------------------------------------------------------------
# value_hidden_activation / policy_hidden_activation (reference utils/common_utils.py:16-45), id = dsact_config value
ACTIVATIONS = {"gelu": (0, nn.GELU), "relu": (1, nn.ReLU), "elu": (2, nn.ELU), "selu": (3, nn.SELU),
               "sigmoid": (4, nn.Sigmoid), "tanh": (5, nn.Tanh)}


# policy_std_type (reference networks/mlp.py:43-73): "mlp_shared" (every example), "parameter", and "mlp_separated" (two MLPs `mean` /
# `log_std` side by side in the arena like the CNN nets' twin trunks: dsact_config.policy_twin)
STD_TYPES = ("mlp_shared", "parameter", "mlp_separated")


# value_output_activation / policy_output_activation (utils/common_utils.py:16-45; the module behind the last Linear,
# networks/mlp.py:15-20): id = dsact_config value (0 linear, else the hidden-activation id; 6 = "gelu", whose derivative needs the
# pre-activation: tile-stage kernels only, round 6)
OUT_ACTIVATIONS = {"linear": (0, nn.Identity), "relu": (1, nn.ReLU), "elu": (2, nn.ELU), "selu": (3, nn.SELU),
                   "sigmoid": (4, nn.Sigmoid), "tanh": (5, nn.Tanh), "gelu": (6, nn.GELU)}


def _mlp(sizes, activation="gelu", out_activation="linear"):
    layers = []
    for j in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[j], sizes[j + 1]))
        layers.append(ACTIVATIONS[activation][1]() if j < len(sizes) - 2 else OUT_ACTIVATIONS[out_activation][1]())
    return nn.Sequential(*layers)


class HipActionValueDistri(nn.Module):
    """Distributional Q(s,a) -> (mean, std); parameter names as reference networks/mlp.py:109-127."""

    def __init__(self, obs_dim, act_dim, hidden, activation="gelu", out_activation="linear"):
        super().__init__()
        self.q = _mlp([obs_dim + act_dim] + list(hidden) + [2], activation, out_activation)

    def forward(self, obs, act):
        out = self.q(torch.cat([obs, act], dim=-1))
        return torch.cat((out[..., :1], nn.functional.softplus(out[..., 1:])), dim=-1)


class HipStochaPolicy(nn.Module):
    """Stochastic policy obs -> (mean | std); parameter names as reference networks/mlp.py:28-100."""

    def __init__(self, obs_dim, act_dim, hidden, act_high, act_low, min_log_std, max_log_std, activation="gelu",
                 std_type="mlp_shared", out_activation="linear"):
        super().__init__()
        self.std_type = std_type
        if std_type == "parameter":   # networks/mlp.py:63-73: the MLP gives the mean, log_std is a learnable parameter
            self.mean = _mlp([obs_dim] + list(hidden) + [act_dim], activation, out_activation)
            self.log_std = nn.Parameter(-0.5 * torch.ones(1, act_dim))
        elif std_type == "mlp_separated":   # networks/mlp.py:46-57: mean and log_std from two MLPs, constructed in this order
            self.mean = _mlp([obs_dim] + list(hidden) + [act_dim], activation, out_activation)
            self.log_std = _mlp([obs_dim] + list(hidden) + [act_dim], activation, out_activation)
        else:
            self.policy = _mlp([obs_dim] + list(hidden) + [2 * act_dim], activation, out_activation)
        self.min_log_std, self.max_log_std = float(min_log_std), float(max_log_std)
        self.register_buffer("act_high_lim", torch.from_numpy(np.asarray(act_high, dtype=np.float32).copy()))
        self.register_buffer("act_low_lim", torch.from_numpy(np.asarray(act_low, dtype=np.float32).copy()))
        self._engine = None  # set by ApproxContainer.attach for the ONLINE policy only

    def forward(self, obs):
        if self._engine is not None:
            # parameters live in the HIP arena: the fused-MLP kernels serve the forward
            self._engine.note_torch_writes(self.parameters())
            lg = self._engine.policy_forward(obs.detach().cpu().numpy())
            return torch.from_numpy(lg).reshape(*obs.shape[:-1], lg.shape[-1]).to(obs.device)
        if self.std_type == "parameter":
            mean = self.mean(obs)
            log_std = self.log_std + torch.zeros_like(mean)
        elif self.std_type == "mlp_separated":
            mean, log_std = self.mean(obs), self.log_std(obs)
        else:
            out = self.policy(obs)
            mean, log_std = torch.chunk(out, chunks=2, dim=-1)
        return torch.cat((mean, torch.clamp(log_std, self.min_log_std, self.max_log_std).exp()), dim=-1)

    action_distribution_cls = TanhGaussDistribution   # reference networks/mlp.py:77; ApproxContainer sets the configured class

    def get_act_dist(self, logits):
        dist = self.action_distribution_cls(logits)
        dist.act_high_lim = self.act_high_lim.to(logits.device)
        dist.act_low_lim = self.act_low_lim.to(logits.device)
        return dist


def _cnn(obs_shape, conv_type):
    """Conv2d + ReLU stack, construction order of reference networks/cnn.py:30-53."""
    _, ks, ch, st, _ = CONV_TYPES[conv_type]
    layers, c = [], int(obs_shape[0])
    for k, co, s in zip(ks, ch, st):
        layers += [nn.Conv2d(c, co, k, s), nn.ReLU()]
        c = co
    return nn.Sequential(*layers)


def _feat_dim(obs_shape, conv_type):
    return CnnArenaLayout(obs_shape, 1, conv_type).feat_dim


class HipCnnActionValueDistri(nn.Module):
    """conv -> flatten -> cat(act) -> separate `mean` / `log_std` MLPs (reference networks/cnn.py:383-461)."""

    def __init__(self, obs_shape, act_dim, conv_type, activation="gelu"):
        super().__init__()
        hidden = CONV_TYPES[conv_type][4]
        self.conv = _cnn(obs_shape, conv_type)
        sizes = [_feat_dim(obs_shape, conv_type) + act_dim] + list(hidden) + [1]
        self.mean = _mlp(sizes, activation)
        self.log_std = _mlp(sizes, activation)

    def forward(self, obs, act):
        img = self.conv(obs)
        feature = torch.cat([img.reshape(img.size(0), -1), act], -1)
        return torch.cat((self.mean(feature), nn.functional.softplus(self.log_std(feature))), dim=-1)


class HipCnnStochaPolicy(nn.Module):
    """conv -> flatten -> separate `mean` / `log_std` MLPs (reference networks/cnn.py:151-240)."""

    def __init__(self, obs_shape, act_dim, conv_type, act_high, act_low, min_log_std, max_log_std, activation="gelu"):
        super().__init__()
        hidden = CONV_TYPES[conv_type][4]
        self.conv = _cnn(obs_shape, conv_type)
        sizes = [_feat_dim(obs_shape, conv_type)] + list(hidden) + [act_dim]
        self.mean = _mlp(sizes, activation)
        self.log_std = _mlp(sizes, activation)
        self.min_log_std, self.max_log_std = float(min_log_std), float(max_log_std)
        self.register_buffer("act_high_lim", torch.from_numpy(np.asarray(act_high, dtype=np.float32).copy()))
        self.register_buffer("act_low_lim", torch.from_numpy(np.asarray(act_low, dtype=np.float32).copy()))
        self._engine = None
        self._obs_ndim = len(obs_shape)

    def forward(self, obs):
        if self._engine is not None:
            lead = obs.shape[:-self._obs_ndim]
            lg = self._engine.policy_forward(obs.detach().cpu().numpy())
            return torch.from_numpy(lg).reshape(*lead, lg.shape[-1]).to(obs.device)
        img = self.conv(obs)
        feature = img.reshape(img.size(0), -1)
        std = torch.clamp(self.log_std(feature), self.min_log_std, self.max_log_std).exp()
        return torch.cat((self.mean(feature), std), dim=-1)

    get_act_dist = HipStochaPolicy.get_act_dist


def _conv_type(kwargs):
    """None for the MLP nets, else the conv type shared by value and policy nets."""
    vt, pt = kwargs.get("value_func_type", "MLP"), kwargs.get("policy_func_type", "MLP")
    if vt == "MLP" and pt == "MLP":
        return None
    if vt != "CNN" or pt != "CNN":
        raise NotImplementedError("DSAC_V2_HIP supports value/policy_func_type MLP+MLP or CNN+CNN (got %s / %s)" % (vt, pt))
    cv, cp = kwargs.get("value_conv_type"), kwargs.get("policy_conv_type")
    if cv != cp or cv not in CONV_TYPES:
        raise NotImplementedError("value_conv_type and policy_conv_type must be the same one of %s (got %s / %s)"
                                  % (sorted(CONV_TYPES), cv, cp))
    return cv


def _hidden_sizes(kwargs):
    ct = _conv_type(kwargs)
    if ct:
        return list(CONV_TYPES[ct][4])
    return list(kwargs["value_hidden_sizes"])   # (policy_hidden_sizes may differ in widths AND depth: _policy_hidden_sizes)


def _policy_hidden_sizes(kwargs):
    """policy_hidden_sizes when they differ from value_hidden_sizes (widths and / or depth: served by the tile-stage kernels), else None"""
    if _conv_type(kwargs):
        return None
    hv, hp = list(kwargs["value_hidden_sizes"]), list(kwargs["policy_hidden_sizes"])
    return hp if hp != hv else None


def _check_supported(kwargs):
    _conv_type(kwargs)
    for key in ("value_hidden_activation", "policy_hidden_activation"):
        if kwargs.get(key, "gelu") not in ACTIVATIONS:
            raise NotImplementedError("DSAC_V2_HIP supports %s in %s (got %r)" % (key, sorted(ACTIVATIONS), kwargs.get(key)))
    for key in ("value_output_activation", "policy_output_activation"):
        got = kwargs.get(key, "linear")
        if got not in OUT_ACTIVATIONS:
            raise NotImplementedError("DSAC_V2_HIP supports %s in %s (got %r)" % (key, sorted(OUT_ACTIVATIONS), got))
        if got != "linear" and _conv_type(kwargs):
            raise NotImplementedError("DSAC_V2_HIP supports %s=%r for the MLP approximators only" % (key, got))
    if kwargs.get("policy_act_distribution", "TanhGaussDistribution") not in ACT_DISTRIBUTIONS:
        raise NotImplementedError("DSAC_V2_HIP supports policy_act_distribution in %s (got %r)"
                                  % (sorted(ACT_DISTRIBUTIONS), kwargs.get("policy_act_distribution")))
    if kwargs.get("cnn_shared", False):
        raise NotImplementedError("cnn_shared is not supported by the HIP path")
    st = kwargs.get("policy_std_type", "mlp_shared")
    if st not in STD_TYPES:
        raise NotImplementedError("DSAC_V2_HIP supports policy_std_type in %s (got %r)" % (sorted(STD_TYPES), st))
    if st != "mlp_shared" and _conv_type(kwargs):
        raise NotImplementedError("policy_std_type=%r is built for the MLP approximators only" % st)


class ApproxContainer(nn.Module):
    """Same members, registration order and state_dict keys as reference dsac_v2.py:19-62.

    Stand-alone it is a plain CPU torch module (samplers / evaluators / PolicyRunner build their own
    copies and load checkpoints into them). `attach(engine)` (done by DSAC_V2_HIP) re-homes every
    parameter as a VIEW into the engine's flat HBM arenas, after which the HIP kernels update the
    storage in place and `state_dict()/load_state_dict()/torch.save` keep working unchanged.
    """

    def __init__(self, **kwargs):
        super().__init__()
        _check_supported(kwargs)
        hidden = _hidden_sizes(kwargs)
        ct = _conv_type(kwargs)
        A = int(kwargs["action_dim"])
        O = tuple(int(v) for v in kwargs["obsv_dim"]) if ct else int(kwargs["obsv_dim"])
        hi = np.asarray(kwargs["action_high_limit"], dtype=np.float32)
        lo = np.asarray(kwargs["action_low_limit"], dtype=np.float32)
        mn = kwargs.get("policy_min_log_std", -20.0)
        mx = kwargs.get("policy_max_log_std", 2.0)
        # construction order == reference, so the same torch seed gives the same initial weights
        va, pa = kwargs.get("value_hidden_activation", "gelu"), kwargs.get("policy_hidden_activation", "gelu")
        if ct:
            self.q1 = HipCnnActionValueDistri(O, A, ct, va)
            self.q2 = HipCnnActionValueDistri(O, A, ct, va)
        else:
            vo = kwargs.get("value_output_activation", "linear")
            self.q1 = HipActionValueDistri(O, A, hidden, va, vo)
            self.q2 = HipActionValueDistri(O, A, hidden, va, vo)
        self.q1_target = copy.deepcopy(self.q1)  # no RNG consumed, like the reference's deepcopy
        self.q2_target = copy.deepcopy(self.q2)
        if ct:
            self.policy = HipCnnStochaPolicy(O, A, ct, hi, lo, mn, mx, pa)
        else:
            self.policy = HipStochaPolicy(O, A, _policy_hidden_sizes(kwargs) or hidden, hi, lo, mn, mx, pa,
                                          kwargs.get("policy_std_type", "mlp_shared"), kwargs.get("policy_output_activation", "linear"))
        self.policy.action_distribution_cls = ACT_DISTRIBUTIONS[kwargs.get("policy_act_distribution", "TanhGaussDistribution")][1]
        self.policy_target = copy.deepcopy(self.policy)
        for net in (self.policy_target, self.q1_target, self.q2_target):
            for p in net.parameters():
                p.requires_grad = False
        self.log_alpha = nn.Parameter(torch.tensor(1, dtype=torch.float32))
        # nn.Module.__setattr__ would register these as sub-state; keep them out of state_dict
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_layout", CnnArenaLayout(O, A, ct) if ct else
                           ArenaLayout(O, A, hidden, policy_std_type=kwargs.get("policy_std_type", "mlp_shared"),
                                       policy_hidden=_policy_hidden_sizes(kwargs)))

    # reference dsac_v2.py:61-62
    def create_action_distributions(self, logits):
        return self.policy.get_act_dist(logits)

    def _named_param_slots(self):
        """[(parameter, arena_name, storage offset, shape, strides)] for every parameter incl. log_alpha."""
        lay = self._layout
        out = []
        for net in lay.all_nets:
            mod = getattr(self, net)
            params = dict(mod.named_parameters())
            for suffix, arena, off, shape, strides in lay.param_views(net):
                out.append((params[suffix], arena, off, shape, strides))
        out.append((self.log_alpha, "online", lay.log_alpha_offset, (), ()))
        return out

    def attach(self, engine: DsactEngine):
        """Move the current parameter values into the engine's arenas and alias them."""
        arenas = {"online": engine.online, "target": engine.target}
        if isinstance(engine.layout, type(self._layout)):
            object.__setattr__(self, "_layout", engine.layout)   # (the engine decides the stored widths: ArenaLayout pad_to)
        with torch.no_grad():
            for p, arena, off, shape, strides in self._named_param_slots():
                # a (possibly strided) window of the flat arena: conv weights are stored [Cout][KH][KW][Cin],
                # the twin MLPs' output layers sit inside one (n_out x 2H) matrix (include/dsact.h)
                view = torch.as_strided(arenas[arena], shape, strides, off)
                assert tuple(view.shape) == tuple(p.shape), (tuple(view.shape), tuple(p.shape))
                view.copy_(p.data.to(view.device))
                p.data = view
            for net in ("policy", "policy_target"):   # policy_std_type "parameter": the hidden half of the output layer
                zr = getattr(self._layout, "zero_rows", lambda _n: None)(net)
                if zr is not None:
                    arenas[zr[0]][zr[1]:zr[1] + zr[2]].zero_()
                for arena, off, shape, strides in getattr(self._layout, "zero_blocks", lambda _n: [])(net):   # "mlp_separated"
                    torch.as_strided(arenas[arena], shape, strides, off).zero_()
            for pol in (self.policy, self.policy_target):
                pol.act_high_lim = pol.act_high_lim.to(engine.device)
                pol.act_low_lim = pol.act_low_lim.to(engine.device)
        # the copies above ran on torch's stream; the engine's kernels run on its own
        torch.cuda.current_stream(engine.device).synchronize()
        object.__setattr__(self, "_engine", engine)
        self.policy._engine = engine
        # the other nets' forward() is plain torch over views of the arenas (evaluation / debugging): let the
        # engine's in-flight update finish first (its kernels run on their own stream)
        for child in self.children():
            if child is not self.policy and not getattr(child, "_dsact_sync_hook", False):
                child.register_forward_pre_hook(lambda _m, _a, _e=engine: _e.sync())
                child._dsact_sync_hook = True
        engine.set_action_limits(self.policy.act_high_lim.cpu().numpy(), self.policy.act_low_lim.cpu().numpy())

    def _apply(self, fn, *a, **k):
        # Attached parameters are views of HIP-owned arenas: `.to()/.cpu()/.cuda()` (e.g. the
        # reference trainer's ModuleOnDevice ping-pong, utils/common_utils.py:164-177) must not
        # re-home them. Acting with CPU observations is served by HipStochaPolicy.forward.
        if self._engine is not None:
            return self
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        if self._engine is not None:
            self._engine.sync()
            state_dict = {k: v.to(self._engine.device) for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._engine is not None:
            torch.cuda.current_stream(self._engine.device).synchronize()   # torch's copies land before the next update
            self._engine.set_action_limits(self.policy.act_high_lim.cpu().numpy(),
                                           self.policy.act_low_lim.cpu().numpy())
            self._engine.policy_dirty()   # the host-side acting snapshot follows the loaded weights
        return out

    def state_dict(self, *a, **k):
        if self._engine is not None:
            self._engine.sync()
        return super().state_dict(*a, **k)


# --------------------------------------------------------------------------------------------------
# lazily materialised tb_info: the only host sync of an update, paid only when somebody reads it
# --------------------------------------------------------------------------------------------------
class LazyTbInfo(dict):
    def __init__(self, alg, serial, alg_time_ms):
        super().__init__()
        self._alg, self._serial, self._done = alg, serial, False
        dict.__setitem__(self, ALG_TIME_KEY, alg_time_ms)

    def _stats(self):
        """the last update: reduced on demand; one of the STATS_SLOTS-1 updates before it: the snapshot the algorithm
        took when it issued the next update (a reference-style caller may log the previous tb_info late)"""
        alg = self._alg
        if alg._serial == self._serial:
            return alg.engine.read_stats()
        if 0 < alg._serial - self._serial < alg.engine.STATS_SLOTS:
            return alg.engine.read_stats(slot=self._serial)
        raise RuntimeError("tb_info of update %d was read %d updates later; device statistics are kept for the last "
                           "%d updates only" % (self._serial, alg._serial - self._serial, alg.engine.STATS_SLOTS))

    def _materialize(self):
        if self._done:
            return
        stats = self._stats()
        for k in STAT_KEYS:
            v = stats[k]
            if k in ("DSAC2/mean_std1", "DSAC2/mean_std2"):
                v = torch.tensor(v)  # 0-dim tensors in the reference (dsac_v2.py:201-202)
            dict.__setitem__(self, k, v)
        t = dict.pop(self, ALG_TIME_KEY)
        # the reference times a synchronous CPU update; here the call returns after the enqueue, so once the statistics
        # are fetched (the sync has been paid) the update's own device time replaces the enqueue time when it is known
        dev = stats.get("_device_ms", -1.0)
        dict.__setitem__(self, ALG_TIME_KEY, dev if dev >= 0.0 and dev > t else t)  # key order of the reference (time last)
        self._done = True

    def __getitem__(self, k):
        if k != ALG_TIME_KEY:
            self._materialize()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._materialize()
        return dict.get(self, k, default)

    def __contains__(self, k):
        return k == ALG_TIME_KEY or k in STAT_KEYS

    def keys(self):
        self._materialize()
        return dict.keys(self)

    def values(self):
        self._materialize()
        return dict.values(self)

    def items(self):
        self._materialize()
        return dict.items(self)

    def __iter__(self):
        self._materialize()
        return dict.__iter__(self)

    def __len__(self):
        return len(STAT_KEYS) + 1


class HipBatch(dict):
    """Token returned by HipReplayBuffer.sample_batch: the minibatch is already staged inside the
    engine, so local_update skips the host round trip. Behaves like the reference's dict of tensors when somebody
    else indexes it (materialised on demand). The engine stages ONE minibatch at a time: a token that is no longer the
    staged one (the caller sampled ahead, or fed another batch in between) re-gathers its own rows by the indices it
    was drawn with before it is trained on or read."""

    def __init__(self, engine, idxs):
        super().__init__()
        self.engine, self.idxs = engine, np.array(idxs, dtype=np.int64, copy=True)
        self.serial = engine.stage_serial
        # ring write position when the rows were sampled: the reference's batch is a COPY taken at sample time
        # (replay_buffer.py:85-90), so a late re-gather must not silently train on rows add_batch has replaced
        self._ptr0, self._added0 = engine.buffer_ptr, engine.rows_added
        self._fill_epoch0 = engine.fill_epoch

    def _overwritten(self):
        cap, written = self.engine.buffer_capacity, self.engine.rows_added - self._added0
        if self.engine.fill_epoch != self._fill_epoch0:
            # buffer_fill_device wrote rows at an arbitrary position (not the ring's append order): which of the sampled
            # rows it replaced is not known here -- the token counts as overwritten entirely
            return int(self.idxs.size)
        if written <= 0 or cap <= 0:
            return 0
        if written >= cap:
            return int(self.idxs.size)
        return int((((self.idxs - self._ptr0) % cap) < written).sum())

    def restage(self):
        if self.serial != self.engine.stage_serial:
            n = self._overwritten()
            if n:
                raise RuntimeError(
                    "HipBatch: %d of the %d sampled ring rows were overwritten by add_batch after sample_batch; the "
                    "token re-gathers by index, so it would train on other transitions than the ones sampled. Use the "
                    "token before adding to the buffer, or sample again." % (n, self.idxs.size))
            self.engine.gather(self.idxs)
            self.serial = self.engine.stage_serial

    def _fill(self):
        if not dict.__len__(self):
            self.restage()
            b = self.engine.read_batch(with_logp=True)
            for k in ("obs", "obs2", "act", "rew", "done", "logp"):
                dict.__setitem__(self, k, torch.from_numpy(b[k]))

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def items(self):
        self._fill()
        return dict.items(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)


class HipBatchGroup:
    """Token returned by HipReplayBuffer.sample_batches(batch_size, n): the index rows of the next n minibatches, drawn with
    the reference's own n `np.random.randint` calls (replay_buffer.py:86) while the ring cannot change between them. The
    rows are gathered on the device by the graph replay DSAC_V2_HIP.local_update_group issues; like a HipBatch it refuses
    to be trained on once add_batch has replaced sampled rows."""

    def __init__(self, engine, idxs):
        self.engine, self.idxs = engine, np.array(idxs, dtype=np.int64, copy=True)
        assert self.idxs.ndim == 2
        self._ptr0, self._added0, self._fill_epoch0 = engine.buffer_ptr, engine.rows_added, engine.fill_epoch

    def __len__(self):
        return int(self.idxs.shape[0])

    def batch(self, j):
        """minibatch j as an ordinary HipBatch token (re-gathers its rows when used)"""
        b = HipBatch(self.engine, self.idxs[j])
        b.serial = -1
        b._ptr0, b._added0, b._fill_epoch0 = self._ptr0, self._added0, self._fill_epoch0
        return b

    def check_fresh(self):
        probe = self.batch(0)
        probe.idxs = self.idxs.reshape(-1)
        n = probe._overwritten()
        if n:
            raise RuntimeError("HipBatchGroup: %d of the %d sampled ring rows were overwritten by add_batch after "
                               "sample_batches; sample again" % (n, self.idxs.size))


class _Hyper:
    """An `adjustable_parameters` entry (dsac_v2.py:92-99): the reference re-reads the attribute on every update,
    so assignment after construction must reach the engine (dsact_set_hyper; drops a captured graph)."""

    def __init__(self, name):
        self.name, self.slot = name, "_hp_" + name

    def __get__(self, obj, cls=None):
        return self if obj is None else obj.__dict__[self.slot]

    def __set__(self, obj, value):
        eng = obj.__dict__.get("engine")
        if eng is not None:
            eng.set_hyper(self.name, value)   # raises DsactError on an invalid value; the attribute keeps the old one
        obj.__dict__[self.slot] = value


class DSAC_V2_HIP:
    """DSAC-T on MI355X. kwargs are the reference's flat dict (dsac_v2.py:27-59,81-90) plus additive
    keys: `replay_batch_size` (minibatch rows), `hip_device` (default 0), `strict_rng` (default
    False: device Philox noise; True: draw the 8 torch.randn tensors of App. A.1 on the host in the
    reference's order and inject them -- bit-identical noise to the reference after the same seed),
    `hip_flags` (DSACT_F_*), `global_batch` (data parallel)."""

    gamma, tau, auto_alpha, alpha, delay_update = (_Hyper(n) for n in ("gamma", "tau", "auto_alpha", "alpha", "delay_update"))
    _tb_cls = LazyTbInfo

    def __init__(self, **kwargs):
        _check_supported(kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.gamma = kwargs["gamma"]
        self.tau = kwargs["tau"]
        self.target_entropy = -kwargs["action_dim"]
        self.auto_alpha = kwargs["auto_alpha"]
        self.alpha = kwargs.get("alpha", 0.2)
        self.delay_update = kwargs["delay_update"]
        self.tau_b = kwargs.get("tau_b", self.tau)
        self.strict_rng = bool(kwargs.get("strict_rng", False))
        self.flags = int(kwargs.get("hip_flags", 0))
        B = int(kwargs["replay_batch_size"])
        ct = _conv_type(kwargs)
        self.engine = DsactEngine(
            tuple(kwargs["obsv_dim"]) if ct else int(kwargs["obsv_dim"]), int(kwargs["action_dim"]), _hidden_sizes(kwargs), B,
            conv_type=ct,
            gamma=self.gamma, tau=self.tau, tau_b=self.tau_b, auto_alpha=bool(self.auto_alpha),
            alpha=float(self.alpha), delay_update=int(self.delay_update),
            lr_q=kwargs["value_learning_rate"], lr_pi=kwargs["policy_learning_rate"],
            lr_alpha=kwargs["alpha_learning_rate"],
            min_log_std=kwargs.get("policy_min_log_std", -20.0), max_log_std=kwargs.get("policy_max_log_std", 2.0),
            global_batch=kwargs.get("global_batch"), device=int(kwargs.get("hip_device", 0)),
            value_act=ACTIVATIONS[kwargs.get("value_hidden_activation", "gelu")][0],
            policy_act=ACTIVATIONS[kwargs.get("policy_hidden_activation", "gelu")][0],
            act_dist=ACT_DISTRIBUTIONS[kwargs.get("policy_act_distribution", "TanhGaussDistribution")][0],
            policy_std_type=kwargs.get("policy_std_type", "mlp_shared"),
            value_out_act=0 if ct else OUT_ACTIVATIONS[kwargs.get("value_output_activation", "linear")][0],
            policy_out_act=0 if ct else OUT_ACTIVATIONS[kwargs.get("policy_output_activation", "linear")][0],
            policy_hidden=_policy_hidden_sizes(kwargs),
            # additive: `hip_pad_widths` (default True) -- ragged / unequal hidden widths of the same depth are stored zero-padded to
            # 64 / 128 / 256 when that puts the update on the row-slice chain kernels (dsact/engine.py, dsact/layout.py pad_to)
            pad_widths=bool(kwargs.get("hip_pad_widths", True)))
        self.networks.attach(self.engine)
        register_engine(self.engine)
        # additive: `hip_host_act` (default True) -- the sampler's / evaluator's batch-1 policy forward runs on the host from a
        # pinned snapshot of the policy net refreshed behind every update that moves it (csrc/dsact_host_act.h); False keeps the
        # one-launch GPU forward (csrc/dsact_act.h)
        if not kwargs.get("hip_host_act", True):
            self.engine.debug_set("host_act", 0)
        if not self.strict_rng:
            seed = kwargs.get("seed") or 0
            self.engine.set_device_rng((int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) % (1 << 63) or 1)
        self._serial = 0
        self._last_tb = None

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "auto_alpha", "alpha", "delay_update")

    # ---- optimiser sidecar (additive: the reference's checkpoints hold `networks.state_dict()` only, trainer.py:148-152, so
    #      a resumed reference run restarts Adam and the mean_std EMA from scratch) --------------------------------------
    SIDECAR_FORMAT = "dsact-optimizer-sidecar/1"

    def optimizer_state_dict(self) -> dict:
        """The optimiser-side state `networks.state_dict()` does not hold: both Adam moment arenas (flat, in the engine's
        arena order), the three Adam step counters, the mean_std EMA; plus the arena signature and the algorithm class they
        belong to. Restoring it (after `networks.load_state_dict`) makes the UPDATES continue bit for bit from the same
        minibatches and noise (tests/test_hip_parity.py::test_optimizer_sidecar_roundtrip_resumes_bitwise). It is not a
        whole-run snapshot: the trainer's iteration counter (hence the delay_update phase and the iteration-keyed device
        noise), the replay ring and the NumPy / torch generator states are the caller's to restore."""
        e = self.engine
        e.sync()
        st = e.get_state()
        return {"format": self.SIDECAR_FORMAT, "algorithm": type(self).__name__,
                "signature": [(k, tuple(v.shape)) for k, v in self.networks.state_dict().items()],
                "adam_m": e.adam_m.detach().cpu().clone(), "adam_v": e.adam_v.detach().cpu().clone(),
                "adam_steps": list(st["adam_steps"]), "mean_std": list(st["mean_std"])}

    def load_optimizer_state_dict(self, sd: dict):
        """restores what optimizer_state_dict() saved (after `networks.load_state_dict`); refuses another layout"""
        if sd.get("format") != self.SIDECAR_FORMAT:
            raise ValueError("not a %s file" % self.SIDECAR_FORMAT)
        if sd.get("algorithm", type(self).__name__) != type(self).__name__:
            raise ValueError("optimizer sidecar was written by %s, this is %s" % (sd.get("algorithm"), type(self).__name__))
        sig = [(k, tuple(v.shape)) for k, v in self.networks.state_dict().items()]
        if [(k, tuple(shape)) for k, shape in sd["signature"]] != sig:
            raise ValueError("optimizer sidecar belongs to another network layout")
        e = self.engine
        if sd["adam_m"].numel() != e.adam_m.numel():
            raise ValueError("optimizer sidecar holds %d floats per moment arena, this engine stores %d: the arenas were laid out "
                             "differently (zero-padded hidden widths on one side only? see the `hip_pad_widths` kwarg)"
                             % (sd["adam_m"].numel(), e.adam_m.numel()))
        e.sync()
        e.adam_m.copy_(sd["adam_m"].to(e.adam_m.device))
        e.adam_v.copy_(sd["adam_v"].to(e.adam_v.device))
        torch.cuda.synchronize(e.device)
        e.set_state(adam_steps=[int(v) for v in sd["adam_steps"]], mean_std=[float(v) for v in sd["mean_std"]])

    @property
    def mean_std1(self):
        return self.engine.get_state()["mean_std"][0]

    @property
    def mean_std2(self):
        return self.engine.get_state()["mean_std"][1]

    # ---- staging -------------------------------------------------------------------------------
    def _stage(self, data):
        if isinstance(data, HipBatch) and data.engine is self.engine:
            data.restage()   # no-op when this token is the staged minibatch (HipReplayBuffer fast path)
            return
        # CPU tensors (reference ReplayBuffer.sample_batch) or the CUDA tensors the reference trainer makes of them
        # (training/trainer.py:72-74); CUDA ones are copied device-to-device
        self.engine.load_batch(data["obs"], data["act"], data["rew"], data["obs2"], data["done"])

    def _draw_noise(self):
        """ONE update's draws from the torch global generator in the reference's order -> (eps_new, eps_2, z5, z6)"""
        B, A = self.engine.batch, self.engine.act_dim
        # the reference's 8 draws, in order; 4 of them are discarded there too
        eps_new, eps_2 = torch.randn(B, A), torch.randn(B, A)
        z = [torch.randn(B) for _ in range(6)]
        return eps_new.numpy(), eps_2.numpy(), z[2].numpy(), z[3].numpy()

    def _noise(self):
        if not self.strict_rng:
            return
        self.engine.set_noise(*self._draw_noise())

    # ---- reference surface ------------------------------------------------------------------------
    def _keep_previous_stats(self):
        """A reference-style caller may read update k's tb_info after issuing update k+1: its statistics are reduced
        into a snapshot slot first -- unless nobody can read them any more (the dict was collected) or they already
        have been (materialised): the common case of a trainer that logs every log_save_interval."""
        prev = self._last_tb() if self._last_tb is not None else None
        if self._serial and prev is not None and not prev._done:
            self.engine.stats_snapshot(self._serial)   # asynchronous; see LazyTbInfo._stats

    def _new_tb(self, t0, n_updates=1):
        import weakref

        self._serial += 1
        tb = self._tb_cls(self, self._serial, (time.time() - t0) * 1000 / n_updates)
        self._last_tb = weakref.ref(tb)
        return tb

    def local_update(self, data: Dict, iteration: int) -> dict:
        t0 = time.time()
        self._keep_previous_stats()
        self._stage(data)
        self._noise()
        self.engine.step(int(iteration), self.flags)
        return self._new_tb(t0)

    def local_update_group(self, group: "HipBatchGroup", iteration: int) -> dict:
        """len(group) consecutive { sample_batch -> local_update } of the reference's loop (training/trainer.py:68-82), for
        iterations `iteration` .. `iteration + len(group) - 1`, as ONE graph replay (dsact_run_group: the pipelined graph
        where the shape allows it). `group` comes from HipReplayBuffer.sample_batches, which drew the index rows with the
        reference's own calls; with `strict_rng` the reference's torch.randn draws of those updates are made here, in its
        order, and travel as the graph's noise table. Returns the tb_info of the LAST update (the one a trainer can log:
        HipOffSerialTrainer ends a group at every iteration it has to log, evaluate or save at)."""
        if not (isinstance(group, HipBatchGroup) and group.engine is self.engine):
            raise TypeError("local_update_group takes the HipBatchGroup this engine's HipReplayBuffer.sample_batches returned")
        t0 = time.time()
        group.check_fresh()
        n = len(group)
        # DSACT_F_SKIP_ACTOR_ON_OFF_ITERS: a graph that skips the discarded policy backward is captured per whole delay_update
        # period (dsact_run_group refuses anything else), but a trainer cuts its groups at log / evaluation / checkpoint
        # iterations, wherever those fall. The misaligned head and tail of such a group are issued one update at a time --
        # what local_update accepts at any iteration -- and the aligned middle as the group replay.
        D = int(self.delay_update)
        if (self.flags & 1) and D > 1 and (iteration % D or n % D):
            head = min(n, (-int(iteration)) % D)
            body = ((n - head) // D) * D
            tb, j = None, 0
            while j < n:
                if j == head and body >= 2:
                    tb = self._run_group_rows(group.idxs[j:j + body], int(iteration) + j, time.time())
                    j += body
                else:
                    tb = self.local_update(group.batch(j), int(iteration) + j)
                    j += 1
            return tb
        return self._run_group_rows(group.idxs, int(iteration), t0)

    def _run_group_rows(self, idxs, iteration, t0):
        self._keep_previous_stats()
        n = int(idxs.shape[0])
        noise = None
        if self.strict_rng:
            noise = np.stack([np.concatenate([a.reshape(-1) for a in self._draw_noise()]) for _ in range(n)]).astype(np.float32)
        self.engine.run_group(iteration, idxs, noise, self.flags)
        return self._new_tb(t0, n)

    def _grad_views(self):
        lay, g = self.engine.layout, self.engine.grads
        out = {}
        for net in lay.online_nets:
            views = []
            for _, _, off, shape, strides in lay.param_views(net):
                views.append(torch.as_strided(g, shape, strides, off))
            out[net] = views
        out["log_alpha"] = g[lay.log_alpha_offset:lay.log_alpha_offset + 1].view(())
        return out

    def get_remote_update_info(self, data: Dict, iteration: int) -> Tuple[dict, dict]:
        t0 = time.time()
        self._keep_previous_stats()
        self._stage(data)
        self._noise()
        self.engine.compute_grads(int(iteration), self.flags)
        self.engine.sync()   # the caller reads the returned gradient tensors with torch ops on torch's stream
        tb = self._new_tb(t0)
        v = self._grad_views()
        info = {"q1_grad": v["q1"], "q2_grad": v["q2"], "policy_grad": v["policy"], "iteration": iteration}
        if self.auto_alpha:
            info["log_alpha_grad"] = v["log_alpha"]
        return tb, info

    def remote_update(self, update_info: dict):
        v = self._grad_views()
        self.engine.sync()
        with torch.no_grad():
            for key, name in (("q1_grad", "q1"), ("q2_grad", "q2"), ("policy_grad", "policy")):
                for dst, src in zip(v[name], update_info[key]):
                    if src.data_ptr() != dst.data_ptr():
                        dst.copy_(src.to(dst.device))
            if self.auto_alpha:
                src = update_info["log_alpha_grad"]
                if src.data_ptr() != v["log_alpha"].data_ptr():
                    v["log_alpha"].copy_(src.to(v["log_alpha"].device))
        torch.cuda.current_stream(self.engine.device).synchronize()
        self.engine.apply_update(int(update_info["iteration"]))
