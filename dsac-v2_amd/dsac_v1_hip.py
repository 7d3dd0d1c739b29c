"""DSAC_V1_HIP -- the reference's DSAC_V1 (dsac_v1.py: ONE distributional critic, fixed TD_bound,
variance-weighted critic pseudo-loss) on the same MI355X kernels as DSAC_V2_HIP.

Discovered like every reference algorithm: `create_alg(algorithm="DSAC_V1_HIP", **kwargs)` imports module
`dsac_v1_hip` and instantiates `DSAC_V1_HIP`; samplers / evaluators re-import `ApproxContainer` from it
(reference utils/initialization.py:48-63, training/off_sampler.py:19-23). Surface of reference dsac_v1.py:17-136:

    .networks   ApproxContainer: q, q_target, policy, policy_target, log_alpha (state_dict keys/order == reference)
    .local_update(data, iteration) -> dict with the 8 tb_info keys of dsac_v1.py:171-180
    .get_remote_update_info / .remote_update ("q_grad", "policy_grad", "log_alpha_grad", "iteration")
    .adjustable_parameters

libdsact.so runs the update with `dsact_config.algo = DSACT_ALGO_DSAC_V1`: the tile stages carry one critic chain
per group instead of two, `k_loss_v1` replaces the DSAC-T loss kernel, everything else (replay, gather, heads,
policy backward, fused Adam/Polyak, graphs, data-parallel halves) is shared. Round 4: equal-width MLP nets run the row-slice
chains and the pipelined graph (one critic = fewer units in the same launches); the CNN approximators
(example_train/dsacv1_cnn_carracing_offasync.py) run the conv kernels + tile stages with four conv stacks instead of six.
"""
__all__ = ["ApproxContainer", "DSAC_V1_HIP"]

import copy
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

import dsac_v2_hip as _v2  # noqa: E402
from dsact.engine import DsactEngine, register_engine  # noqa: E402
from dsact.layout import ArenaLayout  # noqa: E402

V1_KEYS = [  # reference dsac_v1.py:171-180 (order preserved) -> index into dsact_read_stats' 16 floats
    ("DSAC/critic_avg_q-RL iter", 0),
    ("DSAC/critic_avg_std-RL iter", 2),
    ("Loss/Actor loss-RL iter", 6),
    ("DSAC/policy_mean-RL iter", 8),
    ("DSAC/policy_std-RL iter", 9),
    ("DSAC/entropy-RL iter", 10),
    ("DSAC/alpha-RL iter", 11),
]
ALG_TIME_KEY = _v2.ALG_TIME_KEY


def _check_supported(kwargs):
    _v2._check_supported(kwargs)   # MLP nets, or the CNN nets of example_train/dsacv1_cnn_carracing_offasync.py (round 4)
    if not _v2._conv_type(kwargs) and list(kwargs["value_hidden_sizes"]) != list(kwargs["policy_hidden_sizes"]):
        # unequal lists run through the zero-padded storage of the row-slice chains only (dsact/layout.py ArenaLayout pad_to): same
        # depth, widths <= 256, no sigmoid, a batch the chains take -- the tile-stage form of unequal widths is DSAC_V2_HIP's
        hv, hp = list(kwargs["value_hidden_sizes"]), list(kwargs["policy_hidden_sizes"])
        pad = DsactEngine._pad_width(hv, hp, int(kwargs["replay_batch_size"]), kwargs["obsv_dim"], algo="DSAC_V1",
                                     value_act=_v2.ACTIVATIONS[kwargs.get("value_hidden_activation", "gelu")][0],
                                     policy_act=_v2.ACTIVATIONS[kwargs.get("policy_hidden_activation", "gelu")][0])
        if not (kwargs.get("hip_pad_widths", True) and pad and "replay_batch_size" in kwargs):
            raise NotImplementedError("DSAC_V1_HIP takes value_hidden_sizes != policy_hidden_sizes only where the row-slice chains can "
                                      "store them zero-padded (same depth, widths <= 256, batch a multiple of 16); got %s / %s" % (hv, hp))
    for key in ("value_output_activation", "policy_output_activation"):
        if kwargs.get(key, "linear") != "linear":
            raise NotImplementedError("DSAC_V1_HIP supports %s='linear' only (output activations are built for DSAC_V2_HIP)" % key)
    if kwargs.get("policy_std_type", "mlp_shared") != "mlp_shared":
        raise NotImplementedError("DSAC_V1_HIP supports policy_std_type='mlp_shared' only (the learnable-parameter log_std is built "
                                  "for DSAC_V2_HIP)")
    if kwargs.get("policy_act_distribution", "TanhGaussDistribution") != "TanhGaussDistribution":
        raise NotImplementedError("DSAC_V1_HIP supports policy_act_distribution='TanhGaussDistribution' only (the plain Gaussian "
                                  "is pinned against the reference for DSAC_V2_HIP)")


class ApproxContainer(_v2.ApproxContainer):
    """Members and registration order of reference dsac_v1.py:23-43."""

    def __init__(self, **kwargs):
        nn.Module.__init__(self)
        _check_supported(kwargs)
        hidden = _v2._hidden_sizes(kwargs)
        A = int(kwargs["action_dim"])
        hi = np.asarray(kwargs["action_high_limit"], dtype=np.float32)
        lo = np.asarray(kwargs["action_low_limit"], dtype=np.float32)
        va, pa = kwargs.get("value_hidden_activation", "gelu"), kwargs.get("policy_hidden_activation", "gelu")
        mn, mx = kwargs.get("policy_min_log_std", -20.0), kwargs.get("policy_max_log_std", 2.0)
        ct = _v2._conv_type(kwargs)
        if ct:   # networks/cnn.py:151-240,383-461: conv stack -> separate `mean` / `log_std` MLPs per net
            O = tuple(int(v) for v in kwargs["obsv_dim"])
            self.q = _v2.HipCnnActionValueDistri(O, A, ct, va)
            self.q_target = copy.deepcopy(self.q)
            self.policy = _v2.HipCnnStochaPolicy(O, A, ct, hi, lo, mn, mx, pa)
            layout = _v2.CnnArenaLayout(O, A, ct, n_critics=1)
        else:
            O = int(kwargs["obsv_dim"])
            self.q = _v2.HipActionValueDistri(O, A, hidden, va)
            self.q_target = copy.deepcopy(self.q)
            self.policy = _v2.HipStochaPolicy(O, A, _v2._policy_hidden_sizes(kwargs) or hidden, hi, lo, mn, mx, pa)
            layout = ArenaLayout(O, A, hidden, n_critics=1, policy_hidden=_v2._policy_hidden_sizes(kwargs))   # (attach: the engine's, maybe padded)
        self.policy_target = copy.deepcopy(self.policy)
        for net in (self.policy_target, self.q_target):
            for p in net.parameters():
                p.requires_grad = False
        self.log_alpha = nn.Parameter(torch.tensor(1, dtype=torch.float32))
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_layout", layout)


class LazyTbInfoV1(_v2.LazyTbInfo):
    def _materialize(self):
        if self._done:
            return
        stats = self._stats()
        vals = list(stats.values())
        A = self._alg.engine.act_dim
        for k, i in V1_KEYS:
            v = vals[i]
            if i in (8, 9):
                v *= A   # the kernel normalises by B*A; DSAC_V1 reports ONE logits element averaged over B
            dict.__setitem__(self, k, v)
        t = dict.pop(self, ALG_TIME_KEY)
        dev = stats.get("_device_ms", -1.0)
        dict.__setitem__(self, ALG_TIME_KEY, dev if dev >= 0.0 and dev > t else t)
        self._done = True

    def __contains__(self, k):
        return k == ALG_TIME_KEY or k in [n for n, _ in V1_KEYS]

    def __len__(self):
        return len(V1_KEYS) + 1


class DSAC_V1_HIP(_v2.DSAC_V2_HIP):
    """kwargs: the reference's flat dict (dsac_v1.py:68-82) plus the additive HIP keys of DSAC_V2_HIP."""
    _tb_cls = LazyTbInfoV1

    TD_bound = _v2._Hyper("TD_bound")
    bound = _v2._Hyper("bound")   # dsac_v1.py:217 `if self.bound:` -- variance-weighted pseudo-loss, else the Gaussian NLL (:227-228)

    def __init__(self, **kwargs):
        _check_supported(kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.gamma = kwargs["gamma"]
        self.tau = kwargs["tau"]
        self.target_entropy = -kwargs["action_dim"]
        self.auto_alpha = kwargs["auto_alpha"]
        self.alpha = kwargs.get("alpha", 0.2)
        self.TD_bound = kwargs.get("TD_bound", 20)
        self.bound = kwargs.get("bound", True)
        self.delay_update = kwargs["delay_update"]
        self.strict_rng = bool(kwargs.get("strict_rng", False))
        self.flags = int(kwargs.get("hip_flags", 0))
        B = int(kwargs["replay_batch_size"])
        ct = _v2._conv_type(kwargs)
        self.engine = DsactEngine(
            tuple(kwargs["obsv_dim"]) if ct else int(kwargs["obsv_dim"]), int(kwargs["action_dim"]), _v2._hidden_sizes(kwargs), B,
            conv_type=ct,
            gamma=self.gamma, tau=self.tau, auto_alpha=bool(self.auto_alpha), alpha=float(self.alpha),
            delay_update=int(self.delay_update), lr_q=kwargs["value_learning_rate"],
            lr_pi=kwargs["policy_learning_rate"], lr_alpha=kwargs["alpha_learning_rate"],
            min_log_std=kwargs.get("policy_min_log_std", -20.0), max_log_std=kwargs.get("policy_max_log_std", 2.0),
            global_batch=kwargs.get("global_batch"), device=int(kwargs.get("hip_device", 0)),
            algo="DSAC_V1", td_bound=float(self.TD_bound), v1_bound=bool(self.bound),
            value_act=_v2.ACTIVATIONS[kwargs.get("value_hidden_activation", "gelu")][0],
            policy_act=_v2.ACTIVATIONS[kwargs.get("policy_hidden_activation", "gelu")][0],
            policy_hidden=None if ct else _v2._policy_hidden_sizes(kwargs),
            pad_widths=bool(kwargs.get("hip_pad_widths", True)))   # ragged / unequal widths on the chains (see DSAC_V2_HIP)
        if not ct and _v2._policy_hidden_sizes(kwargs) and not self.engine.layout.pad_to:
            raise NotImplementedError("DSAC_V1_HIP: the row-slice chains refused the padded shape of unequal value / policy widths")
        self.networks.attach(self.engine)
        register_engine(self.engine)
        if not kwargs.get("hip_host_act", True):   # (see DSAC_V2_HIP: host-side acting forward on / off)
            self.engine.debug_set("host_act", 0)
        if not self.strict_rng:
            seed = kwargs.get("seed") or 0
            self.engine.set_device_rng((int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) % (1 << 63) or 1)
        self._serial = 0
        self._last_tb = None

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "auto_alpha", "alpha", "TD_bound", "bound", "delay_update")

    def _draw_noise(self):
        B, A = self.engine.batch, self.engine.act_dim
        # the reference's 5 draws in order (dsac_v1.py:148-149,201-206,240): the q_target sample is the only z used
        eps_new, eps_2 = torch.randn(B, A), torch.randn(B, A)
        z = [torch.randn(B) for _ in range(3)]
        return eps_new.numpy(), eps_2.numpy(), z[1].numpy(), z[1].numpy()

    def local_update(self, data: Dict, iteration: int) -> dict:
        t0 = time.time()
        self._keep_previous_stats()
        self._stage(data)
        self._noise()
        self.engine.step(int(iteration), self.flags)
        return self._new_tb(t0)

    def get_remote_update_info(self, data: Dict, iteration: int) -> Tuple[dict, dict]:
        t0 = time.time()
        self._keep_previous_stats()
        self._stage(data)
        self._noise()
        self.engine.compute_grads(int(iteration), self.flags)
        self.engine.sync()   # the caller reads the returned gradient tensors with torch ops on torch's stream
        tb = self._new_tb(t0)
        v = self._grad_views()
        info = {"q_grad": v["q"], "policy_grad": v["policy"], "iteration": iteration}
        if self.auto_alpha:
            info["log_alpha_grad"] = v["log_alpha"]
        return tb, info

    def remote_update(self, update_info: dict):
        v = self._grad_views()
        self.engine.sync()
        with torch.no_grad():
            for key, name in (("q_grad", "q"), ("policy_grad", "policy")):
                for dst, src in zip(v[name], update_info[key]):
                    if src.data_ptr() != dst.data_ptr():
                        dst.copy_(src.to(dst.device))
            if self.auto_alpha:
                src = update_info["log_alpha_grad"]
                if src.data_ptr() != v["log_alpha"].data_ptr():
                    v["log_alpha"].copy_(src.to(v["log_alpha"].device))
        torch.cuda.current_stream(self.engine.device).synchronize()
        self.engine.apply_update(int(update_info["iteration"]))
