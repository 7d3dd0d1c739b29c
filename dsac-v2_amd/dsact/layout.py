"""Flat fp32 arena layout shared by the HIP kernels and the torch `state_dict` views.

online  = q1 | q2 | policy | log_alpha
target  = q1_target | q2_target | policy_target
inside a net: W0 (out x in, row-major == nn.Linear.weight) | b0 | W1 | b1 | ...   (include/dsact.h)

Key names and order follow the reference checkpoints (SURVEY.md App. C;
training/trainer.py:148-152 saves `networks.state_dict()`).
"""
from collections import OrderedDict
from typing import List, Optional, Tuple


def mlp_sizes(in_dim: int, hidden: List[int], out_dim: int) -> List[Tuple[int, int]]:
    dims = [in_dim] + list(hidden) + [out_dim]
    return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]  # (out, in) per Linear


def twin_mlp_views(off: int, in0: int, hid: List[int], nb: int):
    """Two MLPs (`mean`, `log_std`: in0 -> hid -> nb each) laid side by side from float `off` on (include/dsact.h, NetDesc::nblk == 2):
      layer 0        [mean.0.weight ; log_std.0.weight]  (2*H0 x in0) | [mean.0.bias ; log_std.0.bias]
      hidden layer l mean.W (H x Hprev) | log_std.W (H x Hprev) | [b_mean ; b_ls]
      output layer   (2 nb x 2H) = [[w_mean, 0], [0, w_ls]] | [b_mean ; b_ls]      (zero blocks structural)
    -> (views of `mean`, views of `log_std`, [(offset, shape, strides)] of the structural zero blocks, end offset);
    a view is (suffix, offset, shape, strides)."""
    L = len(hid)
    mean, lstd, zeros = [], [], []
    width_in = in0
    for l in range(L + 1):
        if l == 0:
            h0 = hid[0]
            mean.append(("mean.0.weight", off, (h0, in0), (in0, 1)))
            lstd.append(("log_std.0.weight", off + h0 * in0, (h0, in0), (in0, 1)))
            off += 2 * h0 * in0
            mean.append(("mean.0.bias", off, (h0,), (1,)))
            lstd.append(("log_std.0.bias", off + h0, (h0,), (1,)))
            off += 2 * h0
            width_in = h0
        elif l < L:
            h, hp = hid[l], width_in
            mean.append(("mean.%d.weight" % (2 * l), off, (h, hp), (hp, 1)))
            lstd.append(("log_std.%d.weight" % (2 * l), off + h * hp, (h, hp), (hp, 1)))
            off += 2 * h * hp
            mean.append(("mean.%d.bias" % (2 * l), off, (h,), (1,)))
            lstd.append(("log_std.%d.bias" % (2 * l), off + h, (h,), (1,)))
            off += 2 * h
            width_in = h
        else:
            hp = width_in
            mean.append(("mean.%d.weight" % (2 * l), off, (nb, hp), (2 * hp, 1)))
            lstd.append(("log_std.%d.weight" % (2 * l), off + nb * 2 * hp + hp, (nb, hp), (2 * hp, 1)))
            zeros.append((off + hp, (nb, hp), (2 * hp, 1)))
            zeros.append((off + nb * 2 * hp, (nb, hp), (2 * hp, 1)))
            off += 2 * nb * 2 * hp
            mean.append(("mean.%d.bias" % (2 * l), off, (nb,), (1,)))
            lstd.append(("log_std.%d.bias" % (2 * l), off + nb, (nb,), (1,)))
            off += 2 * nb
    return mean, lstd, zeros, off


class ArenaLayout:
    def __init__(self, obs_dim: int, act_dim: int, hidden: List[int], n_critics: int = 2, policy_std_type: str = "mlp_shared",
                 policy_hidden: Optional[List[int]] = None, pad_to: Optional[int] = None):
        """n_critics = 2: DSAC_V2 (q1, q2); 1: DSAC_V1 (a single `q`; online = q | policy | log_alpha).

        policy_std_type "parameter" (reference networks/mlp.py:63-73): the arena keeps the policy's output layer in the
        (2 act_dim x H) shape of "mlp_shared" -- the kernels are the same -- with rows [act_dim, 2 act_dim) of the weight
        STRUCTURALLY ZERO (never exposed, their gradient masked: csrc DwProb::msplit) and the second half of the bias being
        the reference's `log_std` parameter: raw log-std = 0 . h + log_std, d log_std = sum over the batch of d raw.

        policy_std_type "mlp_separated" (networks/mlp.py:46-57): the policy is two MLPs `mean` / `log_std` (obs -> hidden -> act_dim
        each), kept side by side like the CNN nets' twin trunks (`twin_mlp_views`; dsact_config.policy_twin).

        pad_to = W (64 / 128 / 256; round 6): every hidden layer of every net is STORED W wide -- the row-slice chain kernels run one
        width per layer across all their units -- and the reference's tensors are the top-left windows of the stored matrices
        (`param_views` strides). The padding is structurally zero and stays zero: a padded feature has zero weights and bias, so its
        pre-activation is 0 and (for every hidden activation with act(0) = 0, i.e. all but sigmoid) its activation is 0; its outgoing
        weights are 0, so its dZ is 0; hence every gradient element that touches the padding is an exact 0 and Adam / Polyak leave
        the zeros in place. n_q / n_pi / n_online count STORED floats; q_shapes / pi_shapes (the cost model) stay the reference's."""
        self.obs_dim, self.act_dim, self.hidden = int(obs_dim), int(act_dim), [int(h) for h in hidden]
        self.n_critics = int(n_critics)
        self.policy_std_type = policy_std_type
        assert policy_std_type in ("mlp_shared", "parameter", "mlp_separated")
        self.policy_twin = policy_std_type == "mlp_separated"
        # value_hidden_sizes != policy_hidden_sizes (utils/common_utils.py:59-62): `hidden` sizes the critics, `policy_hidden` the policy nets
        self.policy_hidden = [int(h) for h in policy_hidden] if policy_hidden is not None else list(self.hidden)
        self.q_shapes = mlp_sizes(obs_dim + act_dim, self.hidden, 2)
        self.pi_shapes = mlp_sizes(obs_dim, self.policy_hidden, 2 * act_dim)
        self.pad_to = int(pad_to) if pad_to else None
        if self.pad_to:
            assert len(self.policy_hidden) == len(self.hidden) and not self.policy_twin
            assert max(self.hidden + self.policy_hidden) <= self.pad_to
            self.stored_hidden = [self.pad_to] * len(self.hidden)
            self.q_stored = mlp_sizes(obs_dim + act_dim, self.stored_hidden, 2)
            self.pi_stored = mlp_sizes(obs_dim, self.stored_hidden, 2 * act_dim)
        else:
            self.stored_hidden = list(self.hidden)
            self.q_stored, self.pi_stored = self.q_shapes, self.pi_shapes
        self.n_q = sum(o * i + o for o, i in self.q_stored)
        self.n_pi = sum(o * i + o for o, i in self.pi_stored)
        if self.policy_twin:
            self.pi_shapes = mlp_sizes(obs_dim, self.policy_hidden, act_dim)      # one trunk
            self._pi_mean, self._pi_lstd, self._pi_zeros, self.n_pi = twin_mlp_views(0, self.obs_dim, self.policy_hidden, self.act_dim)
        nq = self.n_critics
        self.n_online = nq * self.n_q + self.n_pi + 1
        self.n_target = nq * self.n_q + self.n_pi
        if nq == 2:
            self.net_offset = {  # (arena, offset)
                "q1": ("online", 0), "q2": ("online", self.n_q), "policy": ("online", 2 * self.n_q),
                "q1_target": ("target", 0), "q2_target": ("target", self.n_q),
                "policy_target": ("target", 2 * self.n_q),
            }
        else:
            self.net_offset = {"q": ("online", 0), "policy": ("online", self.n_q),
                               "q_target": ("target", 0), "policy_target": ("target", self.n_q)}
        self.log_alpha_offset = self.n_online - 1

    @property
    def online_nets(self):
        return ("q1", "q2", "policy") if self.n_critics == 2 else ("q", "policy")

    @property
    def all_nets(self):
        """registration order of the reference's ApproxContainer (state_dict order)"""
        if self.n_critics == 2:
            return ("q1", "q2", "q1_target", "q2_target", "policy", "policy_target")
        return ("q", "q_target", "policy", "policy_target")

    def net_shapes(self, net: str):
        """the reference's (out, in) per Linear"""
        return self.pi_shapes if net.startswith("policy") else self.q_shapes

    def stored_shapes(self, net: str):
        """(out, in) per Linear as the arena stores it (== net_shapes unless pad_to)"""
        return self.pi_stored if net.startswith("policy") else self.q_stored

    def param_slices(self, net: str):
        """[(suffix, arena, offset, shape)] e.g. ('q.0.weight', 'online', 0, (256, 393)); shape = the reference's, offset = where the
        tensor starts in the arena (pad_to: the row stride is the STORED input width, see param_views)"""
        arena, off = self.net_offset[net]
        sub = "policy" if net.startswith("policy") else "q"
        std_param = net.startswith("policy") and self.policy_std_type == "parameter"
        out = []
        shapes, stored = self.net_shapes(net), self.stored_shapes(net)
        for j, ((o, i), (so, si)) in enumerate(zip(shapes, stored)):
            if std_param and j == len(shapes) - 1:
                # module order of the reference (own parameters before sub-modules): log_std first, then mean.*
                A = self.act_dim
                out.insert(0, ("log_std", arena, off + so * si + A, (1, A)))
                out.append(("mean.%d.weight" % (2 * j), arena, off, (A, i)))
                out.append(("mean.%d.bias" % (2 * j), arena, off + so * si, (A,)))
                off += so * si + so
                continue
            name = "mean" if std_param else sub
            out.append(("%s.%d.weight" % (name, 2 * j), arena, off, (o, i)))
            off += so * si
            out.append(("%s.%d.bias" % (name, 2 * j), arena, off, (o,)))
            off += so
        return out

    def zero_rows(self, net: str):
        """(arena, offset, count) of the structurally-zero weight rows of a policy net (policy_std_type "parameter"), or None"""
        if not (net.startswith("policy") and self.policy_std_type == "parameter"):
            return None
        arena, off = self.net_offset[net]
        shapes = self.stored_shapes(net)
        for o, i in shapes[:-1]:
            off += o * i + o
        o, i = shapes[-1]
        return arena, off + self.act_dim * i, self.act_dim * i

    def zero_blocks(self, net: str):
        """[(arena, storage_offset, shape, strides)] of the structural zero blocks of a twin-trunk policy net's output layer"""
        if not (net.startswith("policy") and self.policy_twin):
            return []
        arena, base = self.net_offset[net]
        return [(arena, base + off, shape, strides) for off, shape, strides in self._pi_zeros]

    def param_views(self, net: str):
        """[(suffix, arena, storage_offset, shape, strides)] -- contiguous views for the MLP nets (the output layers of a
        twin-trunk policy: windows of one (2 act_dim x 2H) matrix), in the reference's state_dict order."""
        if net.startswith("policy") and self.policy_twin:
            arena, base = self.net_offset[net]
            return [(sfx, arena, base + off, shape, strides) for sfx, off, shape, strides in self._pi_mean + self._pi_lstd]
        out = []
        ld = {}   # row stride of layer j's weight = its STORED input width
        for j, (_, si) in enumerate(self.stored_shapes(net)):
            ld[2 * j] = si
        for suffix, arena, off, shape in self.param_slices(net):
            if len(shape) == 2 and suffix != "log_std":
                strides = (ld[int(suffix.split(".")[-2])], 1)
            else:
                strides = (shape[1], 1) if len(shape) == 2 else (1,)
            out.append((suffix, arena, off, shape, strides))
        return out

    def state_dict_keys(self):
        """OrderedDict key -> shape in the reference's registration order."""
        sd = OrderedDict()
        sd["log_alpha"] = ()
        for net in self.all_nets:
            if net.startswith("policy") and self.policy_twin:
                sd[net + ".act_high_lim"] = (self.act_dim,)
                sd[net + ".act_low_lim"] = (self.act_dim,)
                for suffix, _, _, shape, _ in self.param_views(net):
                    sd[net + "." + suffix] = shape
                continue
            slices = self.param_slices(net)
            if net.startswith("policy"):
                if self.policy_std_type == "parameter":   # a module's own parameters precede its buffers
                    sd[net + ".log_std"] = slices[0][3]
                    slices = slices[1:]
                sd[net + ".act_high_lim"] = (self.act_dim,)
                sd[net + ".act_low_lim"] = (self.act_dim,)
            for suffix, _, _, shape in slices:
                sd[net + "." + suffix] = shape
        return sd

    # ---- algorithmic cost model -------------------------------------------
    def mac_per_sample(self):
        """(forward, backward) multiply-accumulates per sample of one update."""
        q_layers = [(o, i) for o, i in self.q_shapes]
        p_layers = [(o, i) for o, i in self.pi_shapes]
        q_fwd = sum(o * i for o, i in q_layers)
        p_fwd = sum(o * i for o, i in p_layers)
        ntr = 2 if self.policy_twin else 1         # "mlp_separated": two trunks of pi_shapes each
        nq = self.n_critics
        fwd = 2 * ntr * p_fwd + 3 * nq * q_fwd
        # critic: dW all layers + dX of layers >= 1 (per critic)
        q_dw = q_fwd
        q_dx = sum(o * i for o, i in q_layers[1:])
        crit = nq * (q_dw + q_dx)
        # actor through the critics: dX only; first layer only the action columns
        w0 = q_layers[0][0]
        act = nq * (q_dx + w0 * self.act_dim)
        # policy: dW all + dX of layers >= 1
        pol = ntr * (p_fwd + sum(o * i for o, i in p_layers[1:]))
        return fwd, crit + act + pol

    def flop_per_step(self, batch: int) -> float:
        f, b = self.mac_per_sample()
        return 2.0 * (f + b) * batch

    def bytes_per_step(self, batch: int, delay_update: int = 2) -> float:
        O, A = self.obs_dim, self.act_dim
        n_on3 = self.n_critics * self.n_q + self.n_pi
        gather = 4 * batch * (2 * O + A + 2) + 4 * batch
        weights = 4 * (n_on3 + self.n_target)
        adam_q = 24 * self.n_critics * self.n_q
        delayed = (24 * self.n_pi + 8 * n_on3) / float(delay_update)
        return float(gather + weights + adam_q + delayed)


# --------------------------------------------------------------------------------------------------
# CNN approximators (reference networks/cnn.py:151-240,383-461; include/dsact.h "CNN nets")
# --------------------------------------------------------------------------------------------------
CONV_TYPES = {
    # name: (id in dsact_config.conv_type, kernel sizes, channels, strides, hidden sizes of the mean / log_std MLPs)
    "type_1": (1, [8, 4, 3], [32, 64, 64], [4, 2, 1], [512, 256]),
    "type_2": (2, [4, 3, 3, 3, 3, 3], [8, 16, 32, 64, 128, 256], [2, 2, 2, 2, 1, 1], [256, 256, 256]),
}


def conv_geometry(obs_shape, conv_type):
    """[(Cin, H, W, Cout, KS, stride, OH, OW)] per conv layer."""
    _, ks, ch, st, _ = CONV_TYPES[conv_type]
    c, h, w = [int(v) for v in obs_shape]
    out = []
    for k, co, s in zip(ks, ch, st):
        oh, ow = (h - k) // s + 1, (w - k) // s + 1
        if oh < 1 or ow < 1:
            raise ValueError("image %s too small for conv_type %s" % (tuple(obs_shape), conv_type))
        out.append((c, h, w, co, k, s, oh, ow))
        c, h, w = co, oh, ow
    return out


class CnnArenaLayout:
    """Arena layout of the CNN nets; same interface as ArenaLayout plus strided views.

    inside a net: conv_j.weight as [Cout][KH][KW][Cin] | conv_j.bias | ... then the twin MLPs side by side:
      layer 0        [mean.0.weight ; log_std.0.weight]  (2*H0 x in) | [mean.0.bias ; log_std.0.bias]
      hidden layer l mean.W (H x Hprev) | log_std.W (H x Hprev) | [b_mean ; b_ls]
      output layer   (n_out x 2H) = [[w_mean, 0], [0, w_ls]] | [b_mean ; b_ls]      (zero blocks structural)
    """

    def __init__(self, obs_shape, act_dim: int, conv_type: str, n_critics: int = 2):
        """n_critics = 2: DSAC_V2 (q1, q2); 1: DSAC_V1 (a single `q`; online = q | policy | log_alpha)."""
        self.n_critics = int(n_critics)
        self.obs_shape = tuple(int(v) for v in obs_shape)
        self.obs_dim = int(self.obs_shape[0] * self.obs_shape[1] * self.obs_shape[2])
        self.act_dim = int(act_dim)
        self.conv_type = conv_type
        self.conv_id, _, _, _, hidden = CONV_TYPES[conv_type]
        self.hidden = list(hidden)
        self.geom = conv_geometry(self.obs_shape, conv_type)
        last = self.geom[-1]
        self.feat_dim = last[3] * last[6] * last[7]
        self._views = {}
        self.n_q = self._build("q", self.feat_dim + self.act_dim, 1)
        self.n_pi = self._build("policy", self.feat_dim, self.act_dim)
        nq = self.n_critics
        self.n_online = nq * self.n_q + self.n_pi + 1
        self.n_target = nq * self.n_q + self.n_pi
        if nq == 2:
            self.net_offset = {
                "q1": ("online", 0), "q2": ("online", self.n_q), "policy": ("online", 2 * self.n_q),
                "q1_target": ("target", 0), "q2_target": ("target", self.n_q),
                "policy_target": ("target", 2 * self.n_q),
            }
            self.online_nets = ("q1", "q2", "policy")
            self.all_nets = ("q1", "q2", "q1_target", "q2_target", "policy", "policy_target")
        else:   # registration order of dsac_v1.py:26-33
            self.net_offset = {"q": ("online", 0), "policy": ("online", self.n_q),
                               "q_target": ("target", 0), "policy_target": ("target", self.n_q)}
            self.online_nets = ("q", "policy")
            self.all_nets = ("q", "q_target", "policy", "policy_target")
        self.log_alpha_offset = self.n_online - 1

    def _build(self, kind, in0, nb):
        """nb = outputs per trunk (1 for Q: mean / std; A for the policy). Returns the float count."""
        views = []  # (suffix, offset, shape, strides)
        off = 0
        for j, (ci, _, _, co, k, _, _, _) in enumerate(self.geom):
            K = k * k * ci
            views.append(("conv.%d.weight" % (2 * j), off, (co, ci, k, k), (K, 1, k * ci, ci)))
            off += co * K
            views.append(("conv.%d.bias" % (2 * j), off, (co,), (1,)))
            off += co
        mean, lstd, _, off = twin_mlp_views(off, in0, self.hidden, nb)
        self._views[kind] = views + mean + lstd
        return off

    def param_views(self, net: str):
        """[(suffix, arena, storage_offset, shape, strides)] in the reference's state_dict order."""
        arena, base = self.net_offset[net]
        kind = "policy" if net.startswith("policy") else "q"
        return [(sfx, arena, base + off, shape, strides) for sfx, off, shape, strides in self._views[kind]]

    def state_dict_keys(self):
        sd = OrderedDict()
        sd["log_alpha"] = ()
        for net in self.all_nets:
            if net.startswith("policy"):
                sd[net + ".act_high_lim"] = (self.act_dim,)
                sd[net + ".act_low_lim"] = (self.act_dim,)
            for suffix, _, _, shape, _ in self.param_views(net):
                sd[net + "." + suffix] = shape
        return sd

    # ---- algorithmic cost model ---------------------------------------------------------------------
    def conv_mac_per_sample(self):
        return [oh * ow * co * k * k * ci for (ci, _, _, co, k, _, oh, ow) in self.geom]

    def mac_per_sample(self):
        """(forward, backward) MACs per sample of one update. Forward as the reference runs it: 2 policy +
        6 Q passes, each with its own conv stack pass EXCEPT that the q(obs, new_act) passes reuse the conv
        features of q(obs, act) (same net, same image). Backward: conv dW (all layers) + dX (layers >= 1) of
        q1, q2, policy; MLPs as in the MLP layout with two trunks."""
        conv = self.conv_mac_per_sample()
        c_fwd = sum(conv)
        c_bwd = sum(conv) + sum(conv[1:])
        hid, A, F = self.hidden, self.act_dim, self.feat_dim

        def trunk(in0, nb):
            dims = [in0] + hid + [nb]
            return [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]

        q, p = trunk(F + A, 1), trunk(F, A)
        q_fwd, p_fwd = 2 * sum(q), 2 * sum(p)
        fwd = 6 * c_fwd + 2 * p_fwd + 6 * q_fwd
        crit = 2 * (q_fwd + 2 * sum(q[1:]) + 2 * hid[0] * F)          # dW all, dX hidden layers, dFeat
        act = 2 * (2 * sum(q[1:]) + 2 * hid[0] * A)
        pol = p_fwd + 2 * sum(p[1:]) + 2 * hid[0] * F
        return fwd, crit + act + pol + 3 * c_bwd

    def flop_per_step(self, batch: int) -> float:
        f, b = self.mac_per_sample()
        return 2.0 * (f + b) * batch

    def bytes_per_step(self, batch: int, delay_update: int = 2) -> float:
        n_on3 = 2 * self.n_q + self.n_pi
        gather = 4 * batch * (2 * self.obs_dim + self.act_dim + 2) + 4 * batch
        weights = 4 * (n_on3 + self.n_target)
        adam_q = 24 * 2 * self.n_q
        delayed = (24 * self.n_pi + 8 * n_on3) / float(delay_update)
        return float(gather + weights + adam_q + delayed)
