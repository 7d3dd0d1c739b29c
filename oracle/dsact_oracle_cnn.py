"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the DSAC-T update with the
reference's CNN approximators (BASELINE.json configs[3]).

Restates, on top of oracle/dsact_oracle.py (same losses / update, only the networks differ):

    networks/cnn.py:30-53     CNN(): Conv2d(k, stride, no padding) + ReLU stack
    networks/cnn.py:151-240   StochaPolicy: conv -> flatten -> SEPARATE `mean` and `log_std` MLPs
    networks/cnn.py:383-461   ActionValueDistri: conv -> flatten -> cat(act) -> SEPARATE `mean` and
                              `log_std` MLPs (each -> 1), softplus on the second

conv_type: "type_2" = kernels [4,3,3,3,3,3], channels [8,16,32,64,128,256], strides [2,2,2,2,1,1],
MLP hidden [256,256,256]; "type_1" = kernels [8,4,3], channels [32,64,64], strides [4,2,1], hidden
[512,256] (networks/cnn.py:173-228). The conv activation is ReLU regardless of `hidden_activation`
(networks/cnn.py:177,205); the MLPs use the configured hidden activation (GELU in the shipped example,
example_train/dsacv2_cnn_carracing_offasync.py:55-77).

Pinning: tests/test_oracle_vs_reference.py::test_cnn_* runs this next to the unmodified reference
(networks.cnn imported through oracle/ref_loader.py) -- same seed, same minibatch, same torch RNG
stream -- and requires equal losses, gradients and post-update parameters.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from collections import OrderedDict
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from .dsact_oracle import DsactOracle, default_config, mlp_forward

CONV_TYPES = {
    # kernel sizes, channels, strides, MLP hidden sizes      (networks/cnn.py:173-228)
    "type_1": ([8, 4, 3], [32, 64, 64], [4, 2, 1], [512, 256]),
    "type_2": ([4, 3, 3, 3, 3, 3], [8, 16, 32, 64, 128, 256], [2, 2, 2, 2, 1, 1], [256, 256, 256]),
}


def conv_out_hw(h, w, kernels, strides):
    dims = []
    for k, s in zip(kernels, strides):
        h, w = (h - k) // s + 1, (w - k) // s + 1
        dims.append((h, w))
    return dims


def cnn_config(obs_shape=(3, 96, 96), act_dim=3, conv_type="type_2", act_limit=1.0, **over):
    """Hyper-parameters of example_train/dsacv2_cnn_carracing_offasync.py (same optimiser settings as
    the MLP example); `obs_dim` is the (C, H, W) tuple the reference passes as `obsv_dim`."""
    ks, ch, st, hid = CONV_TYPES[conv_type]
    cfg = default_config(0, act_dim, hidden=hid, act_limit=act_limit)
    cfg["obs_dim"] = tuple(int(v) for v in obs_shape)
    cfg["conv_type"] = conv_type
    cfg.update(over)
    return cfg


def _new_conv_params(in_ch, kernels, channels, strides) -> List[torch.Tensor]:
    """nn.Conv2d default init in construction order (networks/cnn.py:41-52)."""
    ps = []
    c = in_ch
    for k, co, s in zip(kernels, channels, strides):
        conv = torch.nn.Conv2d(c, co, k, s)
        ps += [conv.weight.detach().clone(), conv.bias.detach().clone()]
        c = co
    return ps


def _new_linears(sizes):
    ps = []
    for j in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[j], sizes[j + 1])
        ps += [lin.weight.detach().clone(), lin.bias.detach().clone()]
    return ps


def conv_forward(x, params, strides, collect=None, masks=None, kink_log=None):
    """Conv2d + ReLU per layer; returns the flattened features (img.view(B, -1), NCHW order).

    masks (parity tests only): per layer a bool tensor -- the ReLU decisions ANOTHER implementation made on the same
    input. ReLU is not differentiable at 0: a pre-activation within rounding noise of 0 lands on either side depending on
    the summation order, and both subgradients are valid. With `masks` the layer computes z * mask instead of relu(z):
    identical wherever the two implementations agree, the other implementation's (equally valid) choice where they do
    not; every disagreement is logged to kink_log as (layer, count, max |z|) so the caller can require |z| ~ 0 there."""
    h = x
    for j, s in enumerate(strides):
        z = F.conv2d(h, params[2 * j], params[2 * j + 1], stride=s)
        if masks is None:
            h = F.relu(z)
        else:
            m = masks[j]
            flip = m != (z > 0)
            if kink_log is not None and bool(flip.any()):
                kink_log.append((j, int(flip.sum()), float(z.detach()[flip].abs().max())))
            h = z * m.to(z.dtype)
        if collect is not None:
            collect.append(h)
    return h.reshape(h.shape[0], -1)


class DsactCnnOracle(DsactOracle):
    """DSAC_V2 with value_func_type = policy_func_type = "CNN" (cnn_shared False)."""

    def __init__(self, cfg: Dict, state_dict=None):
        self.ks, self.ch, self.st, hid = CONV_TYPES[cfg["conv_type"]]
        assert list(cfg["hidden"]) == list(hid)
        C, H, W = cfg["obs_dim"]
        self.n_conv = len(self.ks)
        oh, ow = conv_out_hw(H, W, self.ks, self.st)[-1]
        self.feat_dim = self.ch[-1] * oh * ow
        # keep_conv: record the conv activations of every network call of the next compute_gradient as
        # (net name, [post-ReLU activation per layer, grads retained]) -- per-layer parity / ReLU-kink checks
        self.keep_conv = False
        self.conv_acts = []
        # parity tests: relu_masks[net] = per-layer ReLU decisions of the implementation under test for the conv stack of
        # `net` on the image batch `mask_input` (see conv_forward); kinks collects what that changed
        self.relu_masks, self.mask_input, self.kinks = {}, None, []
        super().__init__(cfg, state_dict)

    # parameter list of one net: conv (w,b)*n_conv | mean MLP (w,b)*(L+1) | log_std MLP (w,b)*(L+1)
    def _new_net(self, extra_in, n_out):
        cfg = self.cfg
        conv = _new_conv_params(cfg["obs_dim"][0], self.ks, self.ch, self.st)
        sizes = [self.feat_dim + extra_in] + list(cfg["hidden"]) + [n_out]
        mean = _new_linears(sizes)       # networks/cnn.py:224-226 / 447
        log_std = _new_linears(sizes)    # networks/cnn.py:227-229 / 448-450
        return conv + mean + log_std

    def _new_q_params(self):
        return self._new_net(self.cfg["act_dim"], 1)

    def _new_pi_params(self):
        return self._new_net(0, self.cfg["act_dim"])

    def _split(self, params):
        nc = 2 * self.n_conv
        nm = (len(params) - nc) // 2
        return params[:nc], params[nc:nc + nm], params[nc + nm:]

    def _conv(self, obs, params, conv):
        acts = [] if self.keep_conv else None
        net_name = next(n for n in self.NETS if self.p[n] is params)
        masks = self.relu_masks.get(net_name) if obs is self.mask_input else None
        log = [] if masks is not None else None
        feat = conv_forward(obs, conv, self.st, acts, masks, log)
        if log:
            self.kinks += [(net_name,) + e for e in log]
        if acts is not None:
            for a in acts:
                if a.requires_grad:
                    a.retain_grad()
            self.conv_acts.append((net_name, acts))
        return feat

    def _pi(self, obs, params, collect=None):
        """StochaPolicy.forward (networks/cnn.py:232-240)."""
        conv, mean, log_std = self._split(params)
        feat = self._conv(obs, params, conv)
        a_mean = mlp_forward(feat, mean, collect, self.cfg.get("policy_act", "gelu"))
        a_std = torch.clamp(mlp_forward(feat, log_std, None, self.cfg.get("policy_act", "gelu")), self.cfg["min_log_std"], self.cfg["max_log_std"]).exp()
        return torch.cat((a_mean, a_std), dim=-1)

    def _q(self, obs, act, params, collect=None):
        """ActionValueDistri.forward (networks/cnn.py:453-461) -> (mean, std)."""
        conv, mean, log_std = self._split(params)
        feat = torch.cat([self._conv(obs, params, conv), act], -1)
        v_mean = mlp_forward(feat, mean, collect, self.cfg.get("value_act", "gelu"))
        v_std = F.softplus(mlp_forward(feat, log_std, None, self.cfg.get("value_act", "gelu")))
        out = torch.cat((v_mean, v_std), dim=-1)
        return out[..., 0], out[..., -1]

    # ---- checkpoint format: <net>.conv.{0,2,..}.*, <net>.mean.{0,2,..}.*, <net>.log_std.{0,2,..}.* -------
    def _names(self, net):
        names = []
        for j in range(self.n_conv):
            names += ["%s.conv.%d.weight" % (net, 2 * j), "%s.conv.%d.bias" % (net, 2 * j)]
        n_lin = len(self.cfg["hidden"]) + 1
        for sub in ("mean", "log_std"):
            for j in range(n_lin):
                names += ["%s.%s.%d.weight" % (net, sub, 2 * j), "%s.%s.%d.bias" % (net, sub, 2 * j)]
        return names

    def state_dict(self):
        sd = OrderedDict()
        sd["log_alpha"] = self.log_alpha.detach().clone()
        for n in self.NETS:
            if n.startswith("policy"):
                sd[n + ".act_high_lim"] = self.act_high.clone()
                sd[n + ".act_low_lim"] = self.act_low.clone()
            for name, t in zip(self._names(n), self.p[n]):
                sd[name] = t.detach().clone()
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.log_alpha.copy_(sd["log_alpha"])
            for n in self.NETS:
                for name, t in zip(self._names(n), self.p[n]):
                    t.copy_(sd[name])


def synth_image_batch(cfg, batch, seed):
    """Synthetic minibatch of the CNN benchmark recipe: images ~ U[0,1) fp32 (CarRacing-style pixel
    scale), act ~ U(lo, hi), rew ~ N(0,1), done ~ Bernoulli(0.01)."""
    rng = np.random.default_rng(seed)
    C, H, W = cfg["obs_dim"]
    A = cfg["act_dim"]
    lo, hi = np.asarray(cfg["act_low"]), np.asarray(cfg["act_high"])
    d = {
        "obs": rng.random((batch, C, H, W), dtype=np.float32),
        "obs2": rng.random((batch, C, H, W), dtype=np.float32),
        "act": (lo + (hi - lo) * rng.random((batch, A), dtype=np.float32)).astype(np.float32),
        "rew": rng.standard_normal(batch).astype(np.float32),
        "done": (rng.random(batch) < 0.01).astype(np.float32),
        "logp": np.zeros(batch, np.float32),
    }
    return {k: torch.as_tensor(v) for k, v in d.items()}
