"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the DSAC-T off-policy update.

This file is the *oracle* for the MI355X-native hot path. It restates, op for op, what the
reference (Jingliang-Duan/DSAC-v2, mounted read-only at /root/reference) computes in

    dsac_v2.py:150-206   DSAC_V2.__compute_gradient
    dsac_v2.py:208-216   __q_evaluate
    dsac_v2.py:218-290   __compute_loss_q          (the three DSAC-T refinements)
    dsac_v2.py:292-302   __compute_target_q
    dsac_v2.py:304-318   __compute_loss_policy / __compute_loss_alpha
    dsac_v2.py:320-347   __update                  (Adam x4, delayed actor/alpha, Polyak)
    networks/mlp.py:15-20,79-100,122-127           MLP, StochaPolicy.forward, ActionValueDistri.forward
    utils/act_distribution_cls.py:44-54            TanhGaussDistribution.rsample
    training/replay_buffer.py:58-90                ring write + uniform index draw + gather

Pinning: `tests/test_oracle_vs_reference.py` runs this restatement next to the unmodified
reference (imported through oracle/ref_loader.py, only possible where /root/reference is mounted)
on the same seeded nets / minibatch / torch RNG stream and requires equality of every loss, stat,
gradient and post-update parameter; `oracle/make_golden.py` stores reference outputs as fixtures
under tests/golden/ so the same pin holds on the GPU box where the reference is absent.
The reference ships no tests or golden vectors of its own for this path, so
the pin is "reference executed here under torch 2.10.0 / numpy 2.2.6", recorded in every fixture.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (dsac-v2_amd/) never does.

Arithmetic is fp32 on CPU; the third-party arithmetic the reference delegates to (torch
nn.functional.linear/gelu/softplus/huber_loss, autograd, torch.optim.Adam, numpy legacy
RandomState.randint) is called here at the same sites instead of being re-derived.
"""
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-6  # utils/act_distribution_cls.py:3

TB_KEYS = [  # dsac_v2.py:188-204, order preserved
    "DSAC2/critic_avg_q1-RL iter",
    "DSAC2/critic_avg_q2-RL iter",
    "DSAC2/critic_avg_std1-RL iter",
    "DSAC2/critic_avg_std2-RL iter",
    "DSAC2/critic_avg_min_std1-RL iter",
    "DSAC2/critic_avg_min_std2-RL iter",
    "Loss/Actor loss-RL iter",
    "Loss/Critic loss-RL iter",
    "DSAC2/policy_mean-RL iter",
    "DSAC2/policy_std-RL iter",
    "DSAC2/entropy-RL iter",
    "DSAC2/alpha-RL iter",
    "DSAC2/mean_std1",
    "DSAC2/mean_std2",
    "Time/Algorithm time [ms]-RL iter",
]


def default_config(obs_dim, act_dim, hidden=(256, 256, 256), act_limit=0.4, **over):
    """Hyper-parameters of example_train/dsacv2_mlp_mujoco_offserial.py:21-141."""
    cfg = dict(
        obs_dim=int(obs_dim), act_dim=int(act_dim), hidden=list(hidden),
        act_high=np.full((act_dim,), act_limit, dtype=np.float32),
        act_low=np.full((act_dim,), -act_limit, dtype=np.float32),
        gamma=0.99, tau=0.005, tau_b=None, auto_alpha=True, alpha=0.2, delay_update=2,
        lr_q=1e-4, lr_pi=1e-4, lr_alpha=3e-4,
        min_log_std=-20.0, max_log_std=0.5,
    )
    cfg.update(over)
    if cfg["tau_b"] is None:
        cfg["tau_b"] = cfg["tau"]  # dsac_v2.py:90
    return cfg


# ----------------------------------------------------------------------------------------------
# networks (networks/mlp.py) as plain parameter lists
# ----------------------------------------------------------------------------------------------
def _new_mlp_params(sizes: List[int]) -> List[torch.Tensor]:
    """nn.Linear default init, consumed from the torch global RNG in the same order as
    networks/mlp.py:15-20 (`nn.Linear(sizes[j], sizes[j+1])` for j ascending)."""
    ps = []
    for j in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[j], sizes[j + 1])
        ps += [lin.weight.detach().clone(), lin.bias.detach().clone()]
    return ps


# value_hidden_activation / policy_hidden_activation (utils/common_utils.py:16-45: torch modules, default arguments)
ACTIVATIONS = {"gelu": F.gelu, "relu": F.relu, "elu": F.elu, "selu": F.selu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}
SELU_ALPHA, SELU_SCALE = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946


def _act_with_side(z, act, pos):
    """relu / selu with the side of 0 chosen by `pos` (bool tensor) instead of by sign(z). Both activations have a kink
    at 0 (their derivative jumps): a pre-activation within rounding noise of 0 lands on either side depending on the
    summation order, and both subgradients are valid. Parity tests pass the decisions of the implementation under test;
    wherever `pos == (z > 0)` this is the activation itself."""
    if act == "relu":
        return z * pos.to(z.dtype)
    return SELU_SCALE * torch.where(pos, z, SELU_ALPHA * torch.expm1(z))


def mlp_forward(x, params, collect=None, act="gelu", sides=None, kink_log=None, out_act="linear"):
    """Linear-act ... Linear (identity output), networks/mlp.py:15-20. sides (parity tests, relu / selu only): per hidden
    layer the bool tensor `z > 0` as another implementation decided it; disagreements are logged as max |z|."""
    n_lin = len(params) // 2
    h = x
    for j in range(n_lin):
        z = F.linear(h, params[2 * j], params[2 * j + 1])
        if collect is not None:
            collect.append(z)
        if j == n_lin - 1:   # output_activation (networks/mlp.py:18: the module behind the last Linear; "linear" = Identity)
            h = z if out_act == "linear" else ACTIVATIONS[out_act](z)
        elif sides is not None and act in ("relu", "selu"):
            flip = sides[j] != (z > 0)
            if kink_log is not None and bool(flip.any()):
                kink_log.append((j, int(flip.sum()), float(z.detach()[flip].abs().max())))
            h = _act_with_side(z, act, sides[j])
        else:
            h = ACTIVATIONS[act](z)
    return h


def policy_forward(obs, params, cfg, collect=None, sides=None, kink_log=None):
    """StochaPolicy.forward (networks/mlp.py:79-100): std_type "mlp_shared" (one MLP -> mean | log_std, the default of every
    example) or "parameter" (cfg["policy_std_type"]: the MLP gives the mean, log_std is a learnable (1, act_dim) parameter --
    here the LAST element of `params`; networks/mlp.py:63-73,92-97)."""
    if cfg.get("policy_std_type", "mlp_shared") == "mlp_separated":
        # networks/mlp.py:46-57,80-85: two MLPs over the observation; params = the `mean` MLP's tensors, then `log_std`'s. The two
        # trunks are walked layer by layer so that `collect` receives ONE row [z_mean | z_log_std] per layer (the layout the HIP
        # arenas keep); the concatenation and the split back are exact, the values are those of two separate nn.Sequential passes
        # (tests/test_oracle_vs_reference.py pins it bit for bit to the live reference).
        n = len(params) // 2
        act, out_act = cfg.get("policy_act", "gelu"), cfg.get("policy_out_act", "linear")
        hm = hl = obs
        for j in range(n // 2):
            zm = F.linear(hm, params[2 * j], params[2 * j + 1])
            zl = F.linear(hl, params[n + 2 * j], params[n + 2 * j + 1])
            last = j == n // 2 - 1
            if collect is not None:
                zc = torch.cat([zm, zl], dim=-1)
                collect.append(zc)
                zm, zl = (t.contiguous() for t in torch.chunk(zc, chunks=2, dim=-1))   # (the next Linear sees the layout it has in the reference)
            if last:
                hm = zm if out_act == "linear" else ACTIVATIONS[out_act](zm)
                hl = zl if out_act == "linear" else ACTIVATIONS[out_act](zl)
            elif sides is not None and act in ("relu", "selu"):
                sm, sl = torch.chunk(sides[j], chunks=2, dim=-1)
                for z, sd in ((zm, sm), (zl, sl)):
                    flip = sd != (z > 0)
                    if kink_log is not None and bool(flip.any()):
                        kink_log.append((j, int(flip.sum()), float(z.detach()[flip].abs().max())))
                hm, hl = _act_with_side(zm, act, sm), _act_with_side(zl, act, sl)
            else:
                hm, hl = ACTIVATIONS[act](zm), ACTIVATIONS[act](zl)
        mean, log_std = hm, hl
    elif cfg.get("policy_std_type", "mlp_shared") == "parameter":
        mean = mlp_forward(obs, params[:-1], collect, cfg.get("policy_act", "gelu"), sides, kink_log, cfg.get("policy_out_act", "linear"))
        log_std = params[-1] + torch.zeros_like(mean)
    else:
        logits = mlp_forward(obs, params, collect, cfg.get("policy_act", "gelu"), sides, kink_log, cfg.get("policy_out_act", "linear"))
        mean, log_std = torch.chunk(logits, chunks=2, dim=-1)
    std = torch.clamp(log_std, cfg["min_log_std"], cfg["max_log_std"]).exp()
    return torch.cat((mean, std), dim=-1)


def q_forward(obs, act, params, collect=None, hidden_act="gelu", sides=None, kink_log=None, out_act="linear"):
    """ActionValueDistri.forward (networks/mlp.py:122-127) -> (mean, std)."""
    logits = mlp_forward(torch.cat([obs, act], dim=-1), params, collect, hidden_act, sides, kink_log, out_act)
    value_mean, value_std = torch.chunk(logits, chunks=2, dim=-1)
    value_std = F.softplus(value_std)
    out = torch.cat((value_mean, value_std), dim=-1)
    return out[..., 0], out[..., -1]


def tanh_gauss_rsample(logits, eps, act_high, act_low):
    """TanhGaussDistribution.rsample (utils/act_distribution_cls.py:44-54) with the standard
    normal draw `eps` injected (Normal.rsample: loc + eps * scale)."""
    mean, std = torch.chunk(logits, chunks=2, dim=-1)
    action = mean + eps * std
    action_limited = (act_high - act_low) / 2 * torch.tanh(action) + (act_high + act_low) / 2
    # torch.distributions.Normal.log_prob, then Independent(...,1) sums the last dim
    var = std ** 2
    log_scale = std.log()
    lp = -((action - mean) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))
    log_prob = (
        lp.sum(-1)
        - torch.log(1 + EPS - torch.pow(torch.tanh(action), 2)).sum(-1)
        - torch.log((act_high - act_low) / 2).sum(-1)
    )
    return action_limited, log_prob


def gauss_rsample(logits, eps, act_high=None, act_low=None):
    """GaussDistribution.rsample (utils/act_distribution_cls.py:99-102; policy_act_distribution = "GaussDistribution") with
    the standard normal draw `eps` injected: no squashing, the action limits are not applied here."""
    mean, std = torch.chunk(logits, chunks=2, dim=-1)
    action = mean + eps * std
    var = std ** 2
    lp = -((action - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))
    return action, lp.sum(-1)


RSAMPLE = {"TanhGaussDistribution": tanh_gauss_rsample, "GaussDistribution": gauss_rsample}


def seeded_state_dict(template, seed):
    """Initial values that regenerate WITHOUT the reference (the GPU box): every Linear weight / bias of the online
    nets drawn by numpy's default_rng(seed) from torch.nn.Linear's own default range U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    (networks/mlp.py builds plain nn.Linear stacks), in state_dict order; the target nets copy
    their online net (dsac_v2.py:44-52 deep copies); every other entry (log_alpha, action limits) keeps the
    template's value. `template`: a state_dict of the right shapes (the reference's or the HIP container's)."""
    rng = np.random.default_rng(seed)
    out, fan_in = {}, 1
    for k, v in template.items():
        net = k.split(".")[0]
        if net.endswith("_target") or not (k.endswith(".weight") or k.endswith(".bias")):
            continue
        if k.endswith(".weight"):
            fan_in = int(v.shape[1])
        b = 1.0 / np.sqrt(fan_in)
        out[k] = torch.as_tensor(rng.uniform(-b, b, tuple(v.shape)).astype(np.float32))
    full = {}
    for k, v in template.items():
        net = k.split(".")[0]
        src = k.replace(net, net[: -len("_target")], 1) if net.endswith("_target") else k
        full[k] = out[src].clone() if src in out else v.detach().clone()
    return full


def draw_noise(batch, act_dim, generator=None):
    """The 8 draws one __compute_gradient consumes from the torch global generator, in order
    (SURVEY.md App. A.1): eps_new[B,A], eps_2[B,A], z3..z8[B]. z3,z4,z7,z8 are drawn and
    discarded by the reference (dsac_v2.py:230-231,306-307); z5,z6 feed q_next_sample."""
    kw = {} if generator is None else {"generator": generator}
    eps_new = torch.randn(batch, act_dim, **kw)
    eps_2 = torch.randn(batch, act_dim, **kw)
    z = [torch.randn(batch, **kw) for _ in range(6)]
    return {"eps_new": eps_new, "eps_2": eps_2, "z5": z[2], "z6": z[3],
            "z_discarded": [z[0], z[1], z[4], z[5]]}


def pad_stored(ps, W, n_tail=0):
    """[W0, b0, W1, b1, ... (+ n_tail trailing tensors)] of an MLP -> the same list with every hidden layer W wide, zero padded:
    the form the HIP arenas STORE when ragged / unequal widths ride the row-slice chains (dsac-v2_amd/dsact/layout.py, ArenaLayout
    pad_to). Test infrastructure only: the flat views the parity tests compare arena against."""
    ps = list(ps)
    n_lin = (len(ps) - n_tail) // 2
    for j in range(n_lin):
        w, b = ps[2 * j], ps[2 * j + 1]
        rows = W if j < n_lin - 1 else w.shape[0]
        cols = w.shape[1] if j == 0 else W
        wp = torch.zeros(rows, cols, dtype=w.dtype); wp[:w.shape[0], :w.shape[1]] = w
        bp = torch.zeros(rows, dtype=b.dtype); bp[:b.shape[0]] = b
        ps[2 * j], ps[2 * j + 1] = wp, bp
    return ps


class DsactOracle:
    """State + one-step semantics of DSAC_V2 (dsac_v2.py:65-347) on CPU fp32."""

    NETS = ("q1", "q2", "q1_target", "q2_target", "policy", "policy_target")

    def __init__(self, cfg: Dict, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.cfg = cfg
        self.act_high = torch.as_tensor(np.asarray(cfg["act_high"], dtype=np.float32))
        self.act_low = torch.as_tensor(np.asarray(cfg["act_low"], dtype=np.float32))
        # construction order of ApproxContainer.__init__ (dsac_v2.py:31-51)
        q1 = self._new_q_params()
        q2 = self._new_q_params()
        pi = self._new_pi_params()
        self.p = {
            "q1": q1, "q2": q2,
            "q1_target": [t.clone() for t in q1], "q2_target": [t.clone() for t in q2],
            "policy": pi, "policy_target": [t.clone() for t in pi],
        }
        self.log_alpha = torch.tensor(1.0, dtype=torch.float32)  # dsac_v2.py:51
        if state_dict is not None:
            self.load_state_dict(state_dict)
        for n in ("q1", "q2", "policy"):
            for t in self.p[n]:
                t.requires_grad_(True)
        self.log_alpha.requires_grad_(True)
        Adam = torch.optim.Adam  # dsac_v2.py:54-59
        self.opt = {
            "q1": Adam(self.p["q1"], lr=cfg["lr_q"]),
            "q2": Adam(self.p["q2"], lr=cfg["lr_q"]),
            "policy": Adam(self.p["policy"], lr=cfg["lr_pi"]),
            "alpha": Adam([self.log_alpha], lr=cfg["lr_alpha"]),
        }
        self.target_entropy = -cfg["act_dim"]  # dsac_v2.py:84
        self.mean_std1 = -1.0  # dsac_v2.py:88-89 sentinel
        self.mean_std2 = -1.0
        self.inter = {}  # intermediates of the last compute_gradient (per-kernel parity tests)

    # ---- networks: overridden by the CNN oracle (oracle/dsact_oracle_cnn.py) ----------------
    def _new_q_params(self):
        cfg = self.cfg
        return _new_mlp_params([cfg["obs_dim"] + cfg["act_dim"]] + list(cfg["hidden"]) + [2])

    def _new_pi_params(self):
        cfg = self.cfg
        hid = list(cfg.get("policy_hidden") or cfg["hidden"])   # policy_hidden_sizes when they differ from value_hidden_sizes
        if self._std_twin:    # networks/mlp.py:46-57: the `mean` MLP, then the `log_std` MLP (construction order = RNG order)
            return (_new_mlp_params([cfg["obs_dim"]] + hid + [cfg["act_dim"]]) + _new_mlp_params([cfg["obs_dim"]] + hid + [cfg["act_dim"]]))
        if self._std_param:   # networks/mlp.py:63-73: the mean MLP, then log_std = -0.5 (no RNG consumed)
            return (_new_mlp_params([cfg["obs_dim"]] + hid + [cfg["act_dim"]])
                    + [torch.full((1, cfg["act_dim"]), -0.5, dtype=torch.float32)])
        return _new_mlp_params([cfg["obs_dim"]] + hid + [2 * cfg["act_dim"]])

    @property
    def _std_param(self):
        return self.cfg.get("policy_std_type", "mlp_shared") == "parameter"

    @property
    def _std_twin(self):
        return self.cfg.get("policy_std_type", "mlp_shared") == "mlp_separated"

    def _named(self, n):
        """[(reference parameter name under net n, tensor)] in the reference's state_dict order"""
        ps = self.p[n]
        if n.startswith("policy") and self._std_twin:
            h = len(ps) // 2
            out = []
            for sub, part in (("mean", ps[:h]), ("log_std", ps[h:])):
                for j in range(len(part) // 2):
                    out += [("%s.%d.weight" % (sub, 2 * j), part[2 * j]), ("%s.%d.bias" % (sub, 2 * j), part[2 * j + 1])]
            return out
        return None

    # Parity tests with relu / selu hidden activations: act_sides[chain] = per hidden layer the bool tensor `z > 0` as the
    # implementation under test decided it, for the differentiated chains "pi", "q1c", "q2c" (first evaluation of the
    # net in an update: (obs, act)) and "q1p", "q2p" (second: (obs, new_act)); act_kinks collects the disagreements.
    act_sides, act_kinks = None, None

    def _chain_of(self, params):
        for n in ("policy", "q1", "q2"):
            if self.p[n] is params:
                k = self._calls.get(n, 0)
                self._calls[n] = k + 1
                return "pi" if n == "policy" else n + ("c" if k == 0 else "p")
        return None

    def _pi(self, obs, params, collect=None):
        ch = self._chain_of(params) if self.act_sides else None
        log = [] if ch in (self.act_sides or {}) else None
        out = policy_forward(obs, params, self.cfg, collect, (self.act_sides or {}).get(ch), log)
        if log:
            self.act_kinks += [(ch,) + e for e in log]
        return out

    def _q(self, obs, act, params, collect=None):
        ch = self._chain_of(params) if self.act_sides else None
        log = [] if ch in (self.act_sides or {}) else None
        out = q_forward(obs, act, params, collect, self.cfg.get("value_act", "gelu"), (self.act_sides or {}).get(ch), log,
                        self.cfg.get("value_out_act", "linear"))
        if log:
            self.act_kinks += [(ch,) + e for e in log]
        return out

    # ---- checkpoint format (training/trainer.py:148-152) -----------------
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict()
        sd["log_alpha"] = self.log_alpha.detach().clone()
        for n in self.NETS:
            is_pi = n.startswith("policy")
            if is_pi:
                sd[n + ".act_high_lim"] = self.act_high.clone()
                sd[n + ".act_low_lim"] = self.act_low.clone()
            sub = "policy" if is_pi else "q"
            ps = self.p[n]
            if is_pi and self._std_twin:
                for name, t in self._named(n):
                    sd[n + "." + name] = t.detach().clone()
                continue
            if is_pi and self._std_param:   # module parameters precede buffers and sub-modules in a state_dict
                sd.pop(n + ".act_high_lim"); sd.pop(n + ".act_low_lim")
                sd[n + ".log_std"] = ps[-1].detach().clone()
                sd[n + ".act_high_lim"] = self.act_high.clone()
                sd[n + ".act_low_lim"] = self.act_low.clone()
                sub = "mean"
            for j in range(len(ps) // 2):
                sd["%s.%s.%d.weight" % (n, sub, 2 * j)] = ps[2 * j].detach().clone()
                sd["%s.%s.%d.bias" % (n, sub, 2 * j)] = ps[2 * j + 1].detach().clone()
        return sd

    def grad_dict(self):
        """gradients of the online nets under the reference's parameter names (the flat views below are in ARENA order)"""
        out = {"log_alpha": self.log_alpha.grad}
        for n in ("q1", "q2", "policy"):
            ps, sub = self.p[n], ("policy" if n == "policy" else "q")
            if n == "policy" and self._std_twin:
                for name, t in self._named(n):
                    out[n + "." + name] = t.grad
                continue
            if n == "policy" and self._std_param:
                out[n + ".log_std"] = ps[-1].grad
                sub = "mean"
            for j in range(len(ps) // 2):
                out["%s.%s.%d.weight" % (n, sub, 2 * j)] = ps[2 * j].grad
                out["%s.%s.%d.bias" % (n, sub, 2 * j)] = ps[2 * j + 1].grad
        return out

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.log_alpha.copy_(sd["log_alpha"])
            for n in self.NETS:
                sub = "policy" if n.startswith("policy") else "q"
                ps = self.p[n]
                if n.startswith("policy") and self._std_twin:
                    for name, t in self._named(n):
                        t.copy_(sd[n + "." + name])
                    continue
                if n.startswith("policy") and self._std_param:
                    ps[-1].copy_(sd[n + ".log_std"])
                    sub = "mean"
                for j in range(len(ps) // 2):
                    ps[2 * j].copy_(sd["%s.%s.%d.weight" % (n, sub, 2 * j)])
                    ps[2 * j + 1].copy_(sd["%s.%s.%d.bias" % (n, sub, 2 * j)])

    # ---- dsac_v2.py:140-148 ------------------------------------------------------------------
    def _alpha(self) -> float:
        if self.cfg["auto_alpha"]:
            return self.log_alpha.exp().item()
        return self.cfg["alpha"]

    @staticmethod
    def _q_eval(mean, std, z):
        z = torch.clamp(z, -3, 3)  # dsac_v2.py:214
        return mean + torch.mul(z, std)

    # ---- dsac_v2.py:150-206 ------------------------------------------------------------------
    def compute_gradient(self, data: Dict[str, torch.Tensor], noise: Dict, keep: bool = False):
        cfg = self.cfg
        obs, act, rew, obs2, done = data["obs"], data["act"], data["rew"], data["obs2"], data["done"]
        I = {} if keep else None
        self._calls, self.act_kinks = {}, []

        def col():
            return [] if keep else None

        c_pi = col()
        logits = self._pi(obs, self.p["policy"], c_pi)
        logits_mean, logits_std = torch.chunk(logits, chunks=2, dim=-1)
        policy_mean = torch.tanh(logits_mean).mean().item()
        policy_std = logits_std.mean().item()
        rsample = RSAMPLE[cfg.get("act_dist", "TanhGaussDistribution")]
        new_act, new_log_prob = rsample(logits, noise["eps_new"], self.act_high, self.act_low)

        for n in ("q1", "q2"):
            self.opt[n].zero_grad()
        # ---- __compute_loss_q (dsac_v2.py:218-290) ----
        logits_2 = self._pi(obs2, self.p["policy_target"])
        act2, log_prob_act2 = rsample(logits_2, noise["eps_2"], self.act_high, self.act_low)
        c_q1, c_q2 = col(), col()
        q1, q1_std = self._q(obs, act, self.p["q1"], c_q1)
        q2, q2_std = self._q(obs, act, self.p["q2"], c_q2)
        tau_b = cfg["tau_b"]
        m1, m2 = torch.mean(q1_std.detach()), torch.mean(q2_std.detach())
        if getattr(self, "std_mean_override", None) is not None:
            # strict data-parallel restatement: the batch means over the GLOBAL batch,
            # supplied by the test harness after its 2-float all-reduce; not part of the reference
            m1, m2 = self.std_mean_override
        if isinstance(self.mean_std1, float) and self.mean_std1 == -1.0:
            self.mean_std1 = m1
        else:
            self.mean_std1 = (1 - tau_b) * self.mean_std1 + tau_b * m1
        if isinstance(self.mean_std2, float) and self.mean_std2 == -1.0:
            self.mean_std2 = m2
        else:
            self.mean_std2 = (1 - tau_b) * self.mean_std2 + tau_b * m2
        q1_next, q1n_std = self._q(obs2, act2, self.p["q1_target"])
        q2_next, q2n_std = self._q(obs2, act2, self.p["q2_target"])
        q1_next_sample = self._q_eval(q1_next, q1n_std, noise["z5"])
        q2_next_sample = self._q_eval(q2_next, q2n_std, noise["z6"])
        q_next = torch.min(q1_next, q2_next)
        q_next_sample = torch.where(q1_next < q2_next, q1_next_sample, q2_next_sample)
        alpha = self._alpha()

        def target_q(q, q_std):  # dsac_v2.py:292-302
            tq = rew + (1 - done) * cfg["gamma"] * (q_next.detach() - alpha * log_prob_act2.detach())
            tqs = rew + (1 - done) * cfg["gamma"] * (q_next_sample.detach() - alpha * log_prob_act2.detach())
            td_bound = 3 * q_std
            difference = torch.clamp(tqs - q, -td_bound, td_bound)
            return tq.detach(), (q + difference).detach()

        target_q1, target_q1_bound = target_q(q1.detach(), self.mean_std1.detach())
        target_q2, target_q2_bound = target_q(q2.detach(), self.mean_std2.detach())
        q1_std_detach = torch.clamp(q1_std, min=0.0).detach()
        q2_std_detach = torch.clamp(q2_std, min=0.0).detach()
        bias = 0.1
        ratio1 = (torch.pow(self.mean_std1, 2) / (torch.pow(q1_std_detach, 2) + bias)).clamp(min=0.1, max=10)
        ratio2 = (torch.pow(self.mean_std2, 2) / (torch.pow(q2_std_detach, 2) + bias)).clamp(min=0.1, max=10)
        hub = lambda a, b: F.huber_loss(a, b, delta=50, reduction="none")
        q1_loss = torch.mean(ratio1 * (hub(q1, target_q1) + q1_std * (
            q1_std_detach.pow(2) - hub(q1.detach(), target_q1_bound)) / (q1_std_detach + bias)))
        q2_loss = torch.mean(ratio2 * (hub(q2, target_q2) + q2_std * (
            q2_std_detach.pow(2) - hub(q2.detach(), target_q2_bound)) / (q2_std_detach + bias)))
        loss_q = q1_loss + q2_loss
        if keep:
            for zl in c_q1 + c_q2 + c_pi:
                zl.retain_grad()
        loss_q.backward()

        # ---- actor (dsac_v2.py:168-181, 304-310) ----
        for n in ("q1", "q2"):
            for t in self.p[n]:
                t.requires_grad_(False)
        self.opt["policy"].zero_grad()
        c_q1p, c_q2p = col(), col()
        q1_pi, _ = self._q(obs, new_act, self.p["q1"], c_q1p)
        q2_pi, _ = self._q(obs, new_act, self.p["q2"], c_q2p)
        loss_policy = (alpha * new_log_prob - torch.min(q1_pi, q2_pi)).mean()
        entropy = -new_log_prob.detach().mean()
        if keep:
            new_act.retain_grad()
            for zl in c_q1p + c_q2p:
                zl.retain_grad()
        loss_policy.backward()
        for n in ("q1", "q2"):
            for t in self.p[n]:
                t.requires_grad_(True)

        # ---- alpha (dsac_v2.py:183-186, 312-318) ----
        if cfg["auto_alpha"]:
            self.opt["alpha"].zero_grad()
            loss_alpha = -self.log_alpha * (new_log_prob.detach() + self.target_entropy).mean()
            loss_alpha.backward()

        tb = OrderedDict()
        vals = [
            q1.detach().mean().item(), q2.detach().mean().item(),
            q1_std.detach().mean().item(), q2_std.detach().mean().item(),
            q1_std.min().detach().item(), q2_std.min().detach().item(),
            loss_policy.item(), loss_q.item(), policy_mean, policy_std, entropy.item(),
            self._alpha(), float(self.mean_std1), float(self.mean_std2), 0.0,
        ]
        for k, v in zip(TB_KEYS, vals):
            tb[k] = v
        if keep:
            I.update(
                logits=logits.detach(), new_act=new_act.detach(), new_log_prob=new_log_prob.detach(),
                logits_2=logits_2.detach(), act2=act2.detach(), log_prob_act2=log_prob_act2.detach(),
                q1=q1.detach(), q1_std=q1_std.detach(), q2=q2.detach(), q2_std=q2_std.detach(),
                q1_next=q1_next.detach(), q2_next=q2_next.detach(),
                q1n_std=q1n_std.detach(), q2n_std=q2n_std.detach(),
                q1_pi=q1_pi.detach(), q2_pi=q2_pi.detach(),
                target_q1=target_q1, target_q2=target_q2,
                target_q1_bound=target_q1_bound, target_q2_bound=target_q2_bound,
                d_new_act=new_act.grad.detach().clone(),
                z_q1=[z.detach() for z in c_q1], dz_q1=[z.grad.detach().clone() for z in c_q1],
                z_q2=[z.detach() for z in c_q2], dz_q2=[z.grad.detach().clone() for z in c_q2],
                z_pi=[z.detach() for z in c_pi], dz_pi=[z.grad.detach().clone() for z in c_pi],
                z_q1p=[z.detach() for z in c_q1p], dz_q1p=[z.grad.detach().clone() for z in c_q1p],
                z_q2p=[z.detach() for z in c_q2p], dz_q2p=[z.grad.detach().clone() for z in c_q2p],
            )
            self.inter = I
        return tb

    # ---- dsac_v2.py:320-347 ------------------------------------------------------------------
    def update(self, iteration: int):
        self.opt["q1"].step()
        self.opt["q2"].step()
        if iteration % self.cfg["delay_update"] == 0:
            self.opt["policy"].step()
            if self.cfg["auto_alpha"]:
                self.opt["alpha"].step()
            with torch.no_grad():
                polyak = 1 - self.cfg["tau"]
                for n in ("q1", "q2", "policy"):
                    for p, p_targ in zip(self.p[n], self.p[n + "_target"]):
                        p_targ.data.mul_(polyak)
                        p_targ.data.add_((1 - polyak) * p.data)

    def local_update(self, data, noise, iteration: int, keep: bool = False):
        tb = self.compute_gradient(data, noise, keep=keep)
        self.update(iteration)
        return tb

    # ---- flat views in the arena order used by the HIP path: q1 | q2 | policy | log_alpha ---
    # policy_std_type "parameter": the HIP arena keeps the policy's output layer in the (2 act_dim x H) shape of mlp_shared --
    # rows [act_dim, 2 act_dim) of the weight are structurally zero, the second half of the bias IS log_std
    # (dsac-v2_amd/dsact/layout.py); the flat views pad the same way.
    def _arena_tensors(self, n, pick):
        return self.arena_order(n, [pick(t) for t in self.p[n]])

    def arena_order(self, n, ps):
        """per-parameter tensors of net n (in the order of self.p[n]) -> flat pieces in the HIP arena's order"""
        if self.cfg.get("pad_to"):
            ps = pad_stored(ps, self.cfg["pad_to"], 1 if (n.startswith("policy") and self._std_param) else 0)
        if n.startswith("policy") and self._std_twin:
            # "mlp_separated": the arena's twin-trunk layout (dsac-v2_amd/dsact/layout.py, twin_mlp_views): layer 0 [W_mean ; W_ls],
            # hidden layers W_mean | W_ls, output layer the dense [[w_mean, 0], [0, w_ls]]; biases [b_mean ; b_ls]
            h = len(ps) // 2
            m, l = ps[:h], ps[h:]
            out = []
            for j in range(h // 2):
                wm, wl = m[2 * j], l[2 * j]
                if j == h // 2 - 1:
                    z = torch.zeros_like(wm)
                    out.append(torch.cat([torch.cat([wm, z], 1), torch.cat([z, wl], 1)], 0).reshape(-1))
                else:
                    out += [wm.reshape(-1), wl.reshape(-1)]
                out += [m[2 * j + 1].reshape(-1), l[2 * j + 1].reshape(-1)]
            return out
        if not (n.startswith("policy") and self._std_param):
            return [t.reshape(-1) for t in ps]
        w, b, ls = ps[-3], ps[-2], ps[-1]
        return [t.reshape(-1) for t in ps[:-3]] + [w.reshape(-1), torch.zeros(w.numel()), b.reshape(-1), ls.reshape(-1)]

    def flat_params(self):
        ts = [t for n in ("q1", "q2", "policy") for t in self._arena_tensors(n, lambda t: t.detach())]
        return torch.cat(ts + [self.log_alpha.detach().reshape(1)])

    def flat_targets(self):
        ts = [t for n in ("q1_target", "q2_target", "policy_target") for t in self._arena_tensors(n, lambda t: t.detach())]
        return torch.cat(ts)

    def flat_grads(self):
        ts = [t for n in ("q1", "q2", "policy") for t in self._arena_tensors(n, lambda t: t.grad.detach())]
        g_a = self.log_alpha.grad if self.log_alpha.grad is not None else torch.zeros(())
        return torch.cat(ts + [g_a.detach().reshape(1)])


# ----------------------------------------------------------------------------------------------
# replay buffer (training/replay_buffer.py:20-90)
# ----------------------------------------------------------------------------------------------
class ReplayOracle:
    """SoA ring buffer with the reference's ptr/size semantics and index draw."""

    def __init__(self, obs_dim, act_dim, max_size):
        self.max_size = int(max_size)
        self.buf = {
            "obs": np.zeros((max_size, obs_dim), np.float32),
            "obs2": np.zeros((max_size, obs_dim), np.float32),
            "act": np.zeros((max_size, act_dim), np.float32),
            "rew": np.zeros(max_size, np.float32),
            "done": np.zeros(max_size, np.float32),
            "logp": np.zeros(max_size, np.float32),
        }
        self.ptr, self.size = 0, 0

    def store(self, obs, info, act, rew, next_obs, done, logp, next_info):
        b = self.buf
        b["obs"][self.ptr] = obs
        b["obs2"][self.ptr] = next_obs
        b["act"][self.ptr] = act
        b["rew"][self.ptr] = rew
        b["done"][self.ptr] = done
        b["logp"][self.ptr] = logp
        self.ptr = (self.ptr + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def add_batch(self, samples):
        for s in samples:
            self.store(*s)

    def draw_indices(self, batch_size):
        return np.random.randint(0, self.size, size=batch_size)  # replay_buffer.py:86 (global legacy RNG)

    def gather(self, idxs):
        return {k: torch.as_tensor(v[idxs], dtype=torch.float32) for k, v in self.buf.items()}

    def sample_batch(self, batch_size):
        return self.gather(self.draw_indices(batch_size))


# ----------------------------------------------------------------------------------------------
# MT19937 + masked rejection == numpy legacy RandomState.randint
# ----------------------------------------------------------------------------------------------
class MT19937:
    """Pure-Python MT19937 (init_genrand seeding as np.random.seed(int) uses)."""

    def __init__(self, seed: int):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            v = mt[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[i] = v
        self.idx = 0

    def next_u32(self) -> int:
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def randint_legacy(rng: MT19937, n: int, size: int) -> np.ndarray:
    """np.random.randint(0, n, size) on the legacy global state, dtype int64 (n-1 < 2**32)."""
    r = n - 1
    out = np.empty(size, np.int64)
    if r == 0:
        out[:] = 0
        return out
    mask = r
    for s in (1, 2, 4, 8, 16):
        mask |= mask >> s
    for i in range(size):
        while True:
            v = rng.next_u32() & mask
            if v <= r:
                break
        out[i] = v
    return out
