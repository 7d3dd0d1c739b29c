"""Flat fp32 arena layout shared by the HIP kernels and the torch `state_dict` views.

online  = q1 | q2 | policy | log_alpha
target  = q1_target | q2_target | policy_target
inside a net: W0 (out x in, row-major == nn.Linear.weight) | b0 | W1 | b1 | ...   (include/dsact.h)

Key names and order follow the reference checkpoints (SURVEY.md App. C;
training/trainer.py:148-152 saves `networks.state_dict()`).
"""
from collections import OrderedDict
from typing import List, Tuple


def mlp_sizes(in_dim: int, hidden: List[int], out_dim: int) -> List[Tuple[int, int]]:
    dims = [in_dim] + list(hidden) + [out_dim]
    return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]  # (out, in) per Linear


class ArenaLayout:
    def __init__(self, obs_dim: int, act_dim: int, hidden: List[int]):
        self.obs_dim, self.act_dim, self.hidden = int(obs_dim), int(act_dim), [int(h) for h in hidden]
        self.q_shapes = mlp_sizes(obs_dim + act_dim, self.hidden, 2)
        self.pi_shapes = mlp_sizes(obs_dim, self.hidden, 2 * act_dim)
        self.n_q = sum(o * i + o for o, i in self.q_shapes)
        self.n_pi = sum(o * i + o for o, i in self.pi_shapes)
        self.n_online = 2 * self.n_q + self.n_pi + 1
        self.n_target = 2 * self.n_q + self.n_pi
        self.net_offset = {  # (arena, offset)
            "q1": ("online", 0), "q2": ("online", self.n_q), "policy": ("online", 2 * self.n_q),
            "q1_target": ("target", 0), "q2_target": ("target", self.n_q),
            "policy_target": ("target", 2 * self.n_q),
        }
        self.log_alpha_offset = self.n_online - 1

    def net_shapes(self, net: str):
        return self.pi_shapes if net.startswith("policy") else self.q_shapes

    def param_slices(self, net: str):
        """[(suffix, arena, offset, shape)] e.g. ('q.0.weight', 'online', 0, (256, 393))"""
        arena, off = self.net_offset[net]
        sub = "policy" if net.startswith("policy") else "q"
        out = []
        for j, (o, i) in enumerate(self.net_shapes(net)):
            out.append(("%s.%d.weight" % (sub, 2 * j), arena, off, (o, i)))
            off += o * i
            out.append(("%s.%d.bias" % (sub, 2 * j), arena, off, (o,)))
            off += o
        return out

    def state_dict_keys(self):
        """OrderedDict key -> shape in the reference's registration order."""
        sd = OrderedDict()
        sd["log_alpha"] = ()
        for net in ("q1", "q2", "q1_target", "q2_target", "policy", "policy_target"):
            if net.startswith("policy"):
                sd[net + ".act_high_lim"] = (self.act_dim,)
                sd[net + ".act_low_lim"] = (self.act_dim,)
            for suffix, _, _, shape in self.param_slices(net):
                sd[net + "." + suffix] = shape
        return sd

    # ---- algorithmic cost model -------------------------------------------
    def mac_per_sample(self):
        """(forward, backward) multiply-accumulates per sample of one update."""
        q_layers = [(o, i) for o, i in self.q_shapes]
        p_layers = [(o, i) for o, i in self.pi_shapes]
        q_fwd = sum(o * i for o, i in q_layers)
        p_fwd = sum(o * i for o, i in p_layers)
        fwd = 2 * p_fwd + 6 * q_fwd
        # critic: dW all layers + dX of layers >= 1 (x2 nets)
        q_dw = q_fwd
        q_dx = sum(o * i for o, i in q_layers[1:])
        crit = 2 * (q_dw + q_dx)
        # actor through q1,q2: dX only; first layer only the action columns
        w0 = q_layers[0][0]
        act = 2 * (q_dx + w0 * self.act_dim)
        # policy: dW all + dX of layers >= 1
        pol = p_fwd + sum(o * i for o, i in p_layers[1:])
        return fwd, crit + act + pol

    def flop_per_step(self, batch: int) -> float:
        f, b = self.mac_per_sample()
        return 2.0 * (f + b) * batch

    def bytes_per_step(self, batch: int, delay_update: int = 2) -> float:
        O, A = self.obs_dim, self.act_dim
        n_on3 = 2 * self.n_q + self.n_pi
        gather = 4 * batch * (2 * O + A + 2) + 4 * batch
        weights = 4 * (n_on3 + self.n_target)
        adam_q = 24 * 2 * self.n_q
        delayed = (24 * self.n_pi + 8 * n_on3) / float(delay_update)
        return float(gather + weights + adam_q + delayed)
