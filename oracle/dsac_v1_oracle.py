"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the reference's DSAC_V1 update
("DSAC_V1 on the same kernels").

    dsac_v1.py:17-53     ApproxContainer: q, q_target, policy, policy_target, log_alpha, 3 Adam
    dsac_v1.py:140-182   __compute_gradient (5 standard-normal draws per call, see draw_noise_v1)
    dsac_v1.py:184-192   __q_evaluate
    dsac_v1.py:194-229   __compute_loss_q   (bound=True: variance-weighted pseudo-loss; bound=False: Gaussian NLL)
    dsac_v1.py:229-236   __compute_target_q (fixed TD_bound)
    dsac_v1.py:238-253   __compute_loss_policy / __compute_loss_alpha
    dsac_v1.py:255-279   __update

The networks are the same MLP classes as DSAC_V2's (networks/mlp.py); their forward is taken from
oracle/dsact_oracle.py. Pinned against the live reference by tests/test_oracle_vs_reference.py::test_v1_*.

Reference quirks restated as they are (parity is against behaviour): `policy_mean` is tanh of logits[..., 0]
only and `policy_std` is logits[..., 1] -- the mean of action dimension 1 when act_dim >= 2 (dsac_v1.py:145-146).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

from .dsact_oracle import _new_mlp_params, pad_stored, policy_forward, q_forward, tanh_gauss_rsample

V1_TB_KEYS = [  # dsac_v1.py:171-180, order preserved
    "DSAC/critic_avg_q-RL iter",
    "DSAC/critic_avg_std-RL iter",
    "Loss/Actor loss-RL iter",
    "DSAC/policy_mean-RL iter",
    "DSAC/policy_std-RL iter",
    "DSAC/entropy-RL iter",
    "DSAC/alpha-RL iter",
    "Time/Algorithm time [ms]-RL iter",
]


def draw_noise_v1(batch, act_dim):
    """The 5 draws of one DSAC_V1.__compute_gradient in order: eps_new[B,A] (policy rsample), eps_2[B,A]
    (policy_target rsample), then one z[B] per __q_evaluate call: q(obs,act) (discarded), q_target(obs2,act2)
    (USED), q(obs,new_act) (discarded)."""
    eps_new = torch.randn(batch, act_dim)
    eps_2 = torch.randn(batch, act_dim)
    z = [torch.randn(batch) for _ in range(3)]
    return {"eps_new": eps_new, "eps_2": eps_2, "z_t": z[1], "z_discarded": [z[0], z[2]]}


class DsacV1Oracle:
    NETS = ("q", "q_target", "policy", "policy_target")

    def __init__(self, cfg: Dict, state_dict=None):
        self.cfg = cfg
        A = cfg["act_dim"]
        self.act_high = torch.as_tensor(np.asarray(cfg["act_high"], dtype=np.float32))
        self.act_low = torch.as_tensor(np.asarray(cfg["act_low"], dtype=np.float32))
        q = self._new_q_params()                           # dsac_v1.py:26-28
        pi = self._new_pi_params()                         # dsac_v1.py:31-33
        self.p = {"q": q, "q_target": [t.clone() for t in q], "policy": pi, "policy_target": [t.clone() for t in pi]}
        self.log_alpha = torch.tensor(1.0, dtype=torch.float32)
        if state_dict is not None:
            self.load_state_dict(state_dict)
        for n in ("q", "policy"):
            for t in self.p[n]:
                t.requires_grad_(True)
        self.log_alpha.requires_grad_(True)
        Adam = torch.optim.Adam
        self.opt = {"q": Adam(self.p["q"], lr=cfg["lr_q"]), "policy": Adam(self.p["policy"], lr=cfg["lr_pi"]),
                    "alpha": Adam([self.log_alpha], lr=cfg["lr_alpha"])}
        self.target_entropy = -A
        self.TD_bound = cfg.get("TD_bound", 20)            # dsac_v1.py:78
        self.bound = bool(cfg.get("bound", True))          # dsac_v1.py:81

    # ---- the approximators (overridden by the CNN variant, oracle/dsac_v1_oracle_cnn.py) ---------------------------
    def _new_q_params(self):
        cfg = self.cfg
        return _new_mlp_params([cfg["obs_dim"] + cfg["act_dim"]] + list(cfg["hidden"]) + [2])

    def _new_pi_params(self):
        cfg = self.cfg
        hid = list(cfg.get("policy_hidden") or cfg["hidden"])   # policy_hidden_sizes when they differ from value_hidden_sizes
        return _new_mlp_params([cfg["obs_dim"]] + hid + [2 * cfg["act_dim"]])

    def _arena(self, ts):
        """a net's tensors as the HIP arena stores them (zero-padded hidden widths: oracle/dsact_oracle.py pad_stored)"""
        return pad_stored(ts, self.cfg["pad_to"]) if self.cfg.get("pad_to") else ts

    def _pi(self, obs, params):
        return policy_forward(obs, params, self.cfg)

    def _q(self, obs, act, params):
        return q_forward(obs, act, params, None, self.cfg.get("value_act", "gelu"))

    def _names(self, n):
        sub = "policy" if n.startswith("policy") else "q"
        names = []
        for j in range(len(self.p[n]) // 2):
            names += ["%s.%s.%d.weight" % (n, sub, 2 * j), "%s.%s.%d.bias" % (n, sub, 2 * j)]
        return names

    def state_dict(self):
        sd = OrderedDict()
        sd["log_alpha"] = self.log_alpha.detach().clone()         # direct parameters precede sub-modules
        for n in ("q", "q_target", "policy", "policy_target"):    # registration order, dsac_v1.py:26-33
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

    def _alpha(self):
        return self.log_alpha.exp().item() if self.cfg["auto_alpha"] else self.cfg["alpha"]

    def compute_gradient(self, data, noise):
        cfg = self.cfg
        obs, act, rew, obs2, done = data["obs"], data["act"], data["rew"], data["obs2"], data["done"]
        logits = self._pi(obs, self.p["policy"])
        policy_mean = torch.tanh(logits[..., 0]).mean().item()     # dsac_v1.py:145
        policy_std = logits[..., 1].mean().item()                   # dsac_v1.py:146
        new_act, new_log_prob = tanh_gauss_rsample(logits, noise["eps_new"], self.act_high, self.act_low)
        self.opt["q"].zero_grad()
        # ---- __compute_loss_q ----
        logits_2 = self._pi(obs2, self.p["policy_target"])
        act2, log_prob_act2 = tanh_gauss_rsample(logits_2, noise["eps_2"], self.act_high, self.act_low)
        q, q_std = self._q(obs, act, self.p["q"])
        qn_mean, qn_std = self._q(obs2, act2, self.p["q_target"])
        q_next_sample = qn_mean + torch.mul(torch.clamp(noise["z_t"], -3, 3), qn_std)
        alpha = self._alpha()
        target_q = rew + (1 - done) * cfg["gamma"] * (q_next_sample.detach() - alpha * log_prob_act2.detach())
        difference = torch.clamp(target_q - q.detach(), -self.TD_bound, self.TD_bound)
        target_q_bound = (q.detach() + difference).detach()
        target_q = target_q.detach()
        q_std_detach = torch.clamp(q_std, min=0.).detach()
        bias = 0.1
        if self.bound:
            q_loss = torch.mean(
                -(target_q - q).detach() / (torch.pow(q_std_detach, 2) + bias) * q
                - ((torch.pow(q.detach() - target_q_bound, 2) - q_std_detach.pow(2)) / (torch.pow(q_std_detach, 3) + bias)) * q_std)
        else:   # dsac_v1.py:227-228: the plain Gaussian negative log-likelihood of the (unbounded) target
            q_loss = -torch.distributions.Normal(q, q_std).log_prob(target_q).mean()
        q_loss.backward()
        for t in self.p["q"]:
            t.requires_grad_(False)
        self.opt["policy"].zero_grad()
        q_pi, _ = self._q(obs, new_act, self.p["q"])
        loss_policy = (alpha * new_log_prob - q_pi).mean()
        entropy = -new_log_prob.detach().mean()
        loss_policy.backward()
        for t in self.p["q"]:
            t.requires_grad_(True)
        if cfg["auto_alpha"]:
            self.opt["alpha"].zero_grad()
            loss_alpha = -self.log_alpha * (new_log_prob.detach() + self.target_entropy).mean()
            loss_alpha.backward()
        vals = [q.detach().mean().item(), q_std.detach().mean().item(), loss_policy.item(), policy_mean, policy_std,
                entropy.item(), self._alpha(), 0.0]
        return OrderedDict(zip(V1_TB_KEYS, vals))

    def update(self, iteration):
        self.opt["q"].step()
        if iteration % self.cfg["delay_update"] == 0:
            self.opt["policy"].step()
            if self.cfg["auto_alpha"]:
                self.opt["alpha"].step()
            with torch.no_grad():
                polyak = 1 - self.cfg["tau"]
                for n in ("q", "policy"):
                    for p, p_targ in zip(self.p[n], self.p[n + "_target"]):
                        p_targ.data.mul_(polyak)
                        p_targ.data.add_((1 - polyak) * p.data)

    def local_update(self, data, noise, iteration):
        tb = self.compute_gradient(data, noise)
        self.update(iteration)
        return tb

    def flat_params(self):
        ts = [t.reshape(-1) for n in ("q", "policy") for t in self._arena([t.detach() for t in self.p[n]])]
        return torch.cat(ts + [self.log_alpha.detach().reshape(1)])

    def flat_targets(self):
        return torch.cat([t.reshape(-1) for n in ("q_target", "policy_target") for t in self._arena([t.detach() for t in self.p[n]])])

    def flat_grads(self):
        ts = [t.reshape(-1) for n in ("q", "policy") for t in self._arena([t.grad.detach() for t in self.p[n]])]
        g_a = self.log_alpha.grad if self.log_alpha.grad is not None else torch.zeros(())
        return torch.cat(ts + [g_a.detach().reshape(1)])
