"""Parity of the HIP path (through the C-ABI) against the oracle and the reference golden vectors.

Gates (BASELINE.json north_star: "losses / stats within 1e-4 of the reference CPU path"):
  * tb_info statistics and the actor loss: 1e-4 ABSOLUTE (no relative slack); the critic loss (a sum of
    squared TD terms, O(1..100)) 1e-5 RELATIVE
  * gradients: 3e-5 of the largest element of the same net (fp32 summation order differs between the MFMA
    k-ordering and the CPU BLAS; measured ~2e-5 on the policy, ~1e-6 on the critics)
  * parameters after the update: 1e-6 absolute for every element, EXCEPT the enumerated ill-conditioned ones: Adam
    divides by sqrt(v_hat) ~ |g|, so an element whose gradient is within a few ulp-of-the-net's-scale of zero gets
    its 1e-8 gradient difference amplified to a visible step difference. Those elements are counted, must stay
    below 0.2 % of the arena, must each be EXPLAINED by their own measured gradient difference
    (|dp_i| <= 1e-6 + 4 * sum_t lr * |dg_i,t| / (sqrt(v_hat_i,t) + eps), see AdamNoise) and can never exceed the
    2*lr-per-update worst case of a sign flip. Polyak targets: 1e-7 + tau x the parameter bound.
  * replay index draws and gathered rows: bit-exact
"""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import (STEP_CASES, hip_kwargs, humanoid_digest, load_step_case, policy_saturation_budget, step_inputs,
                     synth_batch)
from oracle.dsact_oracle import TB_KEYS, DsactOracle, ReplayOracle, default_config, draw_noise

pytestmark = pytest.mark.gpu

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


class Report:
    def __init__(self, title):
        self.title, self.rows, self.bad = title, [], []

    def cmp(self, name, got, want, atol, rtol=0.0):
        got = np.asarray(got, dtype=np.float64).reshape(-1)
        want = np.asarray(want.detach().cpu().numpy() if torch.is_tensor(want) else want, dtype=np.float64).reshape(-1)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        err = float(np.max(np.abs(got - want))) if got.size else 0.0
        scale = float(np.max(np.abs(want))) if want.size else 0.0
        tol = atol + rtol * scale
        ok = bool(np.isfinite(got).all() and err <= tol)
        self.rows.append((name, err, scale, tol, ok))
        if not ok:
            self.bad.append(name)
        return ok

    def cmp_each(self, name, got, want, tol):
        """element-wise tolerance array (an enumerated budget per element instead of one blanket number)"""
        got = np.asarray(got, dtype=np.float64).reshape(-1)
        want = np.asarray(want.detach().cpu().numpy() if torch.is_tensor(want) else want, dtype=np.float64).reshape(-1)
        tol = np.asarray(tol, dtype=np.float64).reshape(-1)
        assert got.shape == want.shape == tol.shape, (name, got.shape, want.shape, tol.shape)
        err = np.abs(got - want)
        ok = bool(np.isfinite(got).all() and (err <= tol).all())
        worst = int(np.argmax(err / tol)) if err.size else 0
        self.rows.append(("%s [%d of %d over the base tolerance]" % (name, int((err > tol.min()).sum()), err.size),
                          float(err[worst]) if err.size else 0.0, float(np.max(np.abs(want))), float(tol[worst]), ok))
        if not ok:
            self.bad.append(name)
        return ok

    def cmp_params(self, name, got, want, noise, atol, lr_steps):
        """every element within atol, except elements whose own gradient-noise bound (AdamNoise.bound) explains more;
        those are enumerated (count, fraction, worst) and capped at the sign-flip worst case 2 * lr * updates."""
        got = np.asarray(got, dtype=np.float64).reshape(-1)
        want = np.asarray(want.detach().cpu().numpy() if torch.is_tensor(want) else want, dtype=np.float64).reshape(-1)
        assert got.shape == want.shape == noise.shape, (name, got.shape, want.shape, noise.shape)
        err = np.abs(got - want)
        over = err > atol
        n_over = int(over.sum())
        unexplained = int((err > atol + noise).sum())
        worst = float(err.max()) if err.size else 0.0
        ratio = float((err[over] / (atol + noise[over])).max()) if n_over else 0.0
        ok = bool(np.isfinite(got).all() and unexplained == 0 and n_over <= 2e-3 * got.size and worst <= 2.0 * lr_steps + atol)
        self.rows.append(("%s [%d of %d over %.0e, worst/bound %.2f]" % (name, n_over, got.size, atol, ratio),
                          worst, float(np.max(np.abs(want))), atol, ok))
        if not ok:
            self.bad.append(name)
        return ok

    def finish(self):
        lines = ["== %s ==" % self.title]
        for name, err, scale, tol, ok in self.rows:
            lines.append("%-28s err %.3e  scale %.3e  tol %.3e  %s" % (name, err, scale, tol, "ok" if ok else "FAIL"))
        txt = "\n".join(lines)
        print(txt)
        try:
            os.makedirs(os.path.dirname(REPORT), exist_ok=True)
            with open(REPORT, "a") as f:
                f.write(txt + "\n")
        except OSError:
            pass
        assert not self.bad, "parity failures in %s: %s" % (self.title, self.bad)


class AdamNoise:
    """Per-element bound on the parameter difference that the measured gradient difference dg = |g_hip - g_ref| can
    cause through torch.optim.Adam (dsac_v2.py:80-86 optimizers): one update moves p by lr * m_hat / (sqrt(v_hat)+eps);
    perturbing g by dg moves m_hat by <= dg and sqrt(v_hat) by <= dg, and m keeps a perturbation for ~1/(1-beta1)
    updates -- bound = 4 * sum_t lr * dg_t / (sqrt(v_hat_t) + eps), v_hat from the REFERENCE gradients."""

    def __init__(self, segs, b2=0.999, eps=1e-8):
        # segs: [(name, n, lr)] in arena order
        self.segs, self.b2, self.eps = segs, b2, eps
        n = sum(s[1] for s in segs)
        self.v = np.zeros(n)
        self.t = {s[0]: 0 for s in segs}
        self.bound = np.zeros(n)
        self.lr_steps = 0.0

    def step(self, g_ref, g_hip, updated):
        g_ref, g_hip = np.asarray(g_ref, np.float64), np.asarray(g_hip, np.float64)
        off, lr_max = 0, 0.0
        for name, n, lr in self.segs:
            sl = slice(off, off + n)
            off += n
            if name not in updated:
                continue
            self.t[name] += 1
            self.v[sl] = self.b2 * self.v[sl] + (1 - self.b2) * g_ref[sl] ** 2
            vhat = self.v[sl] / (1 - self.b2 ** self.t[name])
            self.bound[sl] += 4.0 * lr * np.abs(g_hip[sl] - g_ref[sl]) / (np.sqrt(vhat) + self.eps)
            lr_max = max(lr_max, lr)
        self.lr_steps += lr_max


def make_pair(O, A, hid, B, act_limit=0.4, seed=0, init=None, **over):
    from dsac_v2_hip import DSAC_V2_HIP

    torch.manual_seed(seed)
    alg = DSAC_V2_HIP(**hip_kwargs(O, A, hid, B, act_limit=act_limit, strict_rng=True, **over))
    if init is not None:
        alg.networks.load_state_dict(init)
    cfg = default_config(O, A, hid, act_limit=act_limit, value_act=over.get("value_hidden_activation", "gelu"),
                         policy_act=over.get("policy_hidden_activation", "gelu"),
                         act_dist=over.get("policy_act_distribution", "TanhGaussDistribution"),
                         policy_std_type=over.get("policy_std_type", "mlp_shared"),
                         value_out_act=over.get("value_output_activation", "linear"), policy_out_act=over.get("policy_output_activation", "linear"),
                         policy_hidden=over.get("policy_hidden_sizes"),
                         **{k: over[k] for k in ("auto_alpha", "alpha", "delay_update") if k in over})
    cfg["pad_to"] = getattr(alg.engine.layout, "pad_to", None)   # stored widths of the HIP arenas: the oracle's FLAT views follow them
    orc = DsactOracle(cfg, state_dict={k: v.cpu() for k, v in alg.networks.state_dict().items()})
    return alg, orc


def gelu_np(z):
    return torch.nn.functional.gelu(z).numpy()


KINKED = ("relu", "selu")     # hidden activations whose derivative jumps at 0


def act_np(z, act):
    from oracle.dsact_oracle import ACTIVATIONS
    return ACTIVATIONS[act](z).numpy()


def hip_act_sides(e, cfg, L, B, Lp=None):
    """the side of 0 the HIP kernels put every hidden pre-activation of the differentiated chains on (read back from the
    stored act'(z) for relu, from the sign of the stored activation for selu), for the oracle's `act_sides` (oracle/dsact_oracle.py:
    at a pre-activation within rounding noise of 0 either subgradient is valid; the reference is evaluated with the
    kernels' choice and every disagreement must be such a kink)"""
    sides = {}
    for ch in ("pi", "q1c", "q2c", "q1p", "q2p"):
        act = cfg["policy_act"] if ch == "pi" else cfg["value_act"]
        if act not in KINKED:
            continue
        per = []
        wid = list((cfg.get("policy_hidden") or cfg["hidden"]) if ch == "pi" else cfg["hidden"])   # (the arena may store them padded)
        if ch == "pi" and cfg.get("policy_std_type") == "mlp_separated":
            wid = [2 * w for w in wid]                       # rows [z_mean | z_log_std]
        for l in range((Lp or L) if ch == "pi" else L):     # (policy_hidden_sizes may be a list of another length)
            if act == "relu":     # act'(z) is 0 / 1
                per.append(torch.as_tensor(e.debug_read("G.%s.%d" % (ch, l)).reshape(B, -1)[:, :wid[l]].copy()) > 0.5)
            else:                 # selu: sign(h) == sign(z) (its derivative below 0 passes through the value it has above 0)
                per.append(torch.as_tensor(e.debug_read("H.%s.%d" % (ch, l)).reshape(B, -1)[:, :wid[l]].copy()) > 0)
        sides[ch] = per
    return sides


def compare_intermediates(rep, alg, orc, L, B, A, Lp=None):
    e, I = alg.engine, orc.inter
    d = lambda n: e.debug_read(n)

    def dl(name, want):
        """a hidden-layer buffer [B x stored width] cut to the reference's width; the padding (ArenaLayout pad_to) must be exact zeros"""
        got = np.asarray(d(name)).reshape(B, -1)
        w = int(np.asarray(want).reshape(B, -1).shape[1])
        assert not got[:, w:].any(), "%s: the padded features are not zero" % name
        return got[:, :w]

    ld = e.debug_read("X0").size // B
    O = e.obs_dim
    rep.cmp("new_act", d("XP").reshape(B, ld)[:, O:O + A], I["new_act"], 2e-6)
    rep.cmp("act2", d("X2").reshape(B, ld)[:, O:O + A], I["act2"], 2e-6)
    # log-prob: 2e-4 absolute, plus what fp32 itself does to the tanh correction log(1 + 1e-6 - t^2) of a SATURATED
    # action: t = tanh(x) is rounded to 6e-8 and t^2 once more, so 1 - t^2 carries ~2.4e-7 of absolute noise in BOTH
    # implementations -- divided by (1 + 1e-6 - t^2), which is ~1e-6 when the action sits on its limit. The budget is
    # computed per row from the oracle's own action (t = a / limit); unsaturated rows keep the plain 2e-4.
    lim = float(alg.networks.policy.act_high_lim.max().cpu())

    def logp_tol(act):
        t2 = (np.asarray(act, dtype=np.float64) / lim) ** 2
        return 2e-4 + (2.4e-7 / (1.0 + 1e-6 - np.minimum(t2, 1.0))).sum(axis=1)

    rep.cmp_each("logp_new", d("logp_new"), I["new_log_prob"], logp_tol(I["new_act"]))
    rep.cmp_each("logp2", d("logp2"), I["log_prob_act2"], logp_tol(I["act2"]))
    mu = d("logits_pi").reshape(B, 2 * A)[:, :A]
    rep.cmp("policy_mean", mu, I["logits"][:, :A], 2e-5)
    for i, (q, s) in enumerate((("q1", "q1_std"), ("q2", "q2_std"))):
        o = d("qout_c%d" % i).reshape(B, 2)
        rep.cmp(q, o[:, 0], I[q], 2e-5)
        rep.cmp(s, torch.nn.functional.softplus(torch.as_tensor(o[:, 1])).numpy(), I[s], 2e-5)
    for i, q in enumerate(("q1_next", "q2_next")):
        rep.cmp(q, d("qout_t%d" % i).reshape(B, 2)[:, 0], I[q], 2e-5)
    for i, q in enumerate(("q1_pi", "q2_pi")):
        rep.cmp(q, d("qout_p%d" % i).reshape(B, 2)[:, 0], I[q], 2e-5)
    for ch, key in (("pi", "z_pi"), ("q1c", "z_q1"), ("q2c", "z_q2"), ("q1p", "z_q1p"), ("q2p", "z_q2p")):
        for l in range((Lp or L) if ch == "pi" else L):
            rep.cmp("H.%s.%d" % (ch, l), dl("H.%s.%d" % (ch, l), I[key][l]), act_np(I[key][l], orc.cfg["policy_act" if ch == "pi" else "value_act"]), 2e-6, 2e-5)
    if orc.cfg.get("act_dist", "TanhGaussDistribution") == "TanhGaussDistribution":
        rep.cmp("d_new_act", d("d_new_act"), I["d_new_act"], 1e-9, 2e-4)
    # (GaussDistribution: the oracle's new_act IS the pre-limit sample x, so its .grad also carries d logp / d x through
    #  log_prob(action); the kernels keep dL/d new_act through the critics only and add the log-prob path in the rsample
    #  backward -- the policy's dZ and gradient rows below compare the sum)
    for ch, key in (("q1c", "dz_q1"), ("q2c", "dz_q2"), ("q1p", "dz_q1p"), ("q2p", "dz_q2p"), ("pi", "dz_pi")):
        for l in range((Lp or L) if ch == "pi" else L):
            rep.cmp("dZ.%s.%d" % (ch, l), dl("dZ.%s.%d" % (ch, l), I[key][l]), I[key][l], 1e-10, 2e-4)


def run_case(title, O, A, hid, B, steps, act_limit=0.4, init=None, golden=None, **over):
    rep = Report(title)
    alg, orc = make_pair(O, A, hid, B, act_limit=act_limit, init=init, **over)
    e = alg.engine
    L = len(hid)
    Lp = len(over.get("policy_hidden_sizes") or hid)
    rng = np.random.default_rng(5)
    lay = e.layout
    cfg = orc.cfg
    noise_b = AdamNoise([("q1", lay.n_q, cfg["lr_q"]), ("q2", lay.n_q, cfg["lr_q"]), ("policy", lay.n_pi, cfg["lr_pi"]),
                         ("log_alpha", 1, cfg["lr_alpha"])])
    tau = cfg["tau"]
    for it in range(steps):
        if golden is not None:
            data, noise = step_inputs(golden, it)
        else:
            data = synth_batch(rng, B, O, A, lim=act_limit, p_done=0.05)
            torch.manual_seed(1000 + it)
            noise = draw_noise(B, A)
        keep = it in (0, steps - 1)
        e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
        e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
        e.compute_grads(it)
        e.sync()  # the engine runs on its own stream; torch reads below are on torch's
        if cfg["value_act"] in KINKED or cfg["policy_act"] in KINKED:
            orc.act_sides = hip_act_sides(e, cfg, L, B, Lp)
        orc_chk = orc
        if (cfg["value_act"], cfg["policy_act"]) != ("gelu", "gelu") and it == steps - 1 and it > 0:
            # The intermediates and gradients of the last step are a per-kernel check at tight tolerances, so THEY are taken
            # from a copy of the oracle evaluated AT the parameters the engine holds: after two updates the two trajectories
            # differ by the (enumerated, bounded) Adam noise of the steps before, which the tanh / sigmoid nets' O(1)
            # activations turn into 3e-5 of H -- above gates sized for a summation-order difference. The trajectory oracle
            # itself is NEVER re-synchronised: it takes this step on its own parameters (with its own activation sides) and
            # the params / targets / mean_std rows below compare the accumulated multi-step trajectories.
            orc_chk = copy.deepcopy(orc)
            orc_chk.load_state_dict({k_: v_.cpu() for k_, v_ in alg.networks.state_dict().items()})
            orc.act_sides = None
            orc.compute_gradient(data, noise, keep=False)
        tb_ref = orc_chk.compute_gradient(data, noise, keep=keep)
        for ch, j, cnt, zmax in (orc_chk.act_kinks or []):
            assert zmax < 1e-5, "activation sides differ at a pre-activation of %g (%s layer %d): not a kink" % (zmax, ch, j)
        if keep:
            compare_intermediates(rep, alg, orc_chk, L, B, A, Lp)
        g = e.grads.cpu().numpy()
        g_ref = orc_chk.flat_grads().numpy()
        off = 0
        for net, n in (("q1", lay.n_q), ("q2", lay.n_q), ("policy", lay.n_pi), ("log_alpha", 1)):
            if net == "policy":   # 3e-5 of the tensor's scale + the enumerated fp32 budget of saturated actions
                want = g_ref[off:off + n]
                base = 1e-9 + 3e-5 * float(np.abs(want).max())
                rep.cmp_each("it%d grad.policy" % it, g[off:off + n], want, base + policy_saturation_budget(orc_chk, data, noise, B))
            else:
                rep.cmp("it%d grad.%s" % (it, net), g[off:off + n], g_ref[off:off + n], 1e-9, 3e-5)
            off += n
        if orc_chk is not orc:
            g_ref = orc.flat_grads().numpy()   # the Adam-noise bound follows the TRAJECTORY oracle's gradients
        delayed = it % cfg["delay_update"] == 0
        noise_b.step(g_ref, g, ("q1", "q2") + (("policy",) + (("log_alpha",) if cfg["auto_alpha"] else ()) if delayed else ()))
        e.apply_update(it)
        orc.update(it)
        st = e.read_stats()  # synchronous
        for k in TB_KEYS[:-1]:
            want = float(tb_ref[k])
            if k.startswith("Loss/Critic"):
                rep.cmp("it%d %s" % (it, k.split("/")[-1][:18]), [st[k]], [want], 1e-6, 1e-5)
            else:
                rep.cmp("it%d %s" % (it, k.split("/")[-1][:18]), [st[k]], [want], 1e-4)
        p_hip, t_hip = e.online.cpu().numpy(), e.target.cpu().numpy()
        nb = noise_b.bound
        n_t = t_hip.size
        if golden is not None:
            # the same numbers straight from the unmodified reference
            tb_g = np.asarray(golden["s%d/tb" % it], np.float64)
            crit = TB_KEYS.index("Loss/Critic loss-RL iter")
            keep_i = [i for i in range(len(TB_KEYS) - 1) if i != crit]
            rep.cmp("it%d tb vs reference" % it, [st[TB_KEYS[i]] for i in keep_i], tb_g[keep_i], 1e-4)
            rep.cmp("it%d critic loss vs reference" % it, [st[TB_KEYS[crit]]], [tb_g[crit]], 1e-6, 1e-5)
            rep.cmp_params("it%d params vs reference" % it, p_hip, golden["s%d/params" % it], nb, 1e-6, noise_b.lr_steps)
            rep.cmp_params("it%d targets vs reference" % it, t_hip, golden["s%d/targets" % it], tau * (it + 1) * nb[:n_t], 1e-7,
                           tau * (it + 1) * noise_b.lr_steps)
        rep.cmp_params("it%d params" % it, p_hip, orc.flat_params(), nb, 1e-6, noise_b.lr_steps)
        rep.cmp_params("it%d targets" % it, t_hip, orc.flat_targets(), tau * (it + 1) * nb[:n_t], 1e-7, tau * (it + 1) * noise_b.lr_steps)
        state = e.get_state()
        rep.cmp("it%d mean_std" % it, state["mean_std"], [float(orc.mean_std1), float(orc.mean_std2)], 1e-5, 1e-5)
    # Adam moments of q1 (first parameter tensor) against torch.optim.Adam's state
    st0 = orc.opt["q1"].state[orc.p["q1"][0]]
    n0 = orc.p["q1"][0].numel()
    rep.cmp("adam_m q1.W0", e.adam_m.cpu().numpy()[:n0], st0["exp_avg"].reshape(-1), 1e-10, 1e-3)
    rep.cmp("adam_v q1.W0", e.adam_v.cpu().numpy()[:n0], st0["exp_avg_sq"].reshape(-1), 1e-14, 1e-3)
    assert e.get_state()["adam_steps"][0] == steps
    rep.finish()


def test_humanoid_l3_b256():
    run_case("humanoid 3x256 B=256", 376, 17, (256, 256, 256), 256, steps=4)


def test_humanoid_l2_b256():
    run_case("humanoid 2x256 B=256", 376, 17, (256, 256), 256, steps=3)


def test_ragged_shapes():
    # widths not multiples of the 32x32 tile, batch not a multiple of 4, one action dim
    run_case("ragged O=11 A=3 (96,40) B=50", 11, 3, (96, 40), 50, steps=3)
    run_case("ragged O=5 A=1 (33,) B=7", 5, 1, (33,), 7, steps=3, act_limit=2.0)
    # batch > 448 but neither a multiple of 64 nor of 256: no 64x64 tiles, no split-K -- the generic long-K tile path
    run_case("ragged O=11 A=3 (96,40) B=600", 11, 3, (96, 40), 600, steps=2)
    # multiple of 256 with widths that are not multiples of 64: split-K weight gradients on the 32x32 stage tiles
    run_case("ragged O=11 A=3 (96,40) B=768", 11, 3, (96, 40), 768, steps=2)


def test_large_batch_and_width():
    run_case("B=1024 hidden 512x2", 24, 6, (512, 512), 1024, steps=2)


def test_humanoid_b512_large_batch_tiles():
    """batch >= 512 switches the forward / hidden-backward stages to 64x64 tiles (k_stage64), including the Q nets'
    first layer whose K = 393 is served from the zero-padded operands."""
    run_case("humanoid 3x256 B=512 (64x64 stage tiles)", 376, 17, (256, 256, 256), 512, steps=3)


def test_humanoid_b2048_split_k_weight_gradients():
    """batch > 448 (multiple of 256): the weight gradients are split over 256-sample chunks of the batch (one
    partial gradient arena per chunk, k_sum_parts, then the streaming Adam/Polyak kernel) instead of one
    batch-long contraction per tile."""
    run_case("humanoid 3x256 B=2048 (split-K weight gradients)", 376, 17, (256, 256, 256), 2048, steps=2)


@pytest.mark.parametrize("va,pa", [("relu", "tanh"), ("elu", "selu"), ("sigmoid", "relu"), ("tanh", "elu"), ("selu", "sigmoid")])
def test_hidden_activations(va, pa):
    """value_hidden_activation / policy_hidden_activation other than the examples' gelu (reference
    utils/common_utils.py:16-45): the activation lives in the forward epilogues only (h and act'(z) are both stored) --
    the row-slice chains at the BASELINE shape, the tile path on a ragged shape, the acting forward."""
    run_case("humanoid 3x256 B=256 %s/%s" % (va, pa), 376, 17, (256, 256, 256), 256, steps=3,
             value_hidden_activation=va, policy_hidden_activation=pa)
    run_case("ragged O=11 A=3 (96,40) B=50 %s/%s" % (va, pa), 11, 3, (96, 40), 50, steps=2,
             value_hidden_activation=va, policy_hidden_activation=pa)
    alg, orc = make_pair(24, 6, (64, 64), 16, seed=3, value_hidden_activation=va, policy_hidden_activation=pa)
    from oracle.dsact_oracle import policy_forward
    obs = np.random.default_rng(0).standard_normal((3, 24)).astype(np.float32)
    want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
    got = np.concatenate([alg.engine.policy_forward(obs[i:i + 1]) for i in range(3)])
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(alg.engine.policy_forward(obs), want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,hid,env", [
    (1024, (256, 256, 256), {}),                              # 16-row workgroups (default choice at this batch)
    (1024, (256, 256, 256), {"DSACT_FAT_RT": "2"}),           # 32-row workgroups forced
    (512, (128, 128), {"DSACT_FAT_MIN": "512"}),              # two waves per workgroup (hidden width 128), two layers
    (4096, (256, 256, 256), {}),                              # 32-row workgroups by choice, split-K weight gradients x16
    (1024, (256, 256, 256), {"DSACT_NO_FAT_STAGE": "1"}),     # first layer reading its rows from global memory (rounds 3-5)
])
def test_throughput_regime_kernels(B, hid, env, monkeypatch):
    """dsact_fat.h (batch >= 1024): v_mfma_f32_16x16x4 slices of 16 / 32 rows, style-16 packs of every layer, in-place
    LDS activations -- every intermediate, gradient, statistic and parameter against the oracle, same gates."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    run_case("fat B=%d hidden %s %s" % (B, "x".join(map(str, hid)), env), 376, 17, hid, B, steps=2)
    for k in env:
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("O,A,hid,B,env", [(376, 17, (256, 256, 256), 1024, {}), (17, 6, (256, 256, 256), 1024, {"DSACT_FAT_RT": "2"}),
                                           (11, 3, (128, 128), 512, {"DSACT_FAT_MIN": "512"}), (105, 8, (256, 256, 256), 4096, {})])
def test_throughput_regime_staged_input_rows_equal_global_reads(O, A, hid, B, env, monkeypatch):
    """round 6: the throughput-regime forward copies a slice's input rows into LDS once (chunk layout, zero quads outside a
    segment) and runs its first layer through the hidden layers' loop == the first layer reading the rows from global memory
    chunk by chunk (DSACT_NO_FAT_STAGE=1), bit for bit: same operands, same MFMA order -- observation widths that are and are
    not multiples of 4 / 16, one and two action chunks, 16- and 32-row workgroups, eager updates and a graph replay."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    algs = []
    # default (staged rows), rows from global memory
    for switch in (None, "DSACT_NO_FAT_STAGE"):
        if switch:
            monkeypatch.setenv(switch, "1")
        alg, _ = make_pair(O, A, hid, B, seed=23)
        assert alg.engine.chain_active
        algs.append(alg)
        if switch:
            monkeypatch.delenv(switch, raising=False)
    rng = np.random.default_rng(14)
    for it in range(3):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(500 + it)
        noise = draw_noise(B, A)
        for a in algs:
            a.engine.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            a.engine.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
            a.engine.step(it)
    N = 8192
    for a in algs:
        e = a.engine
        e.set_device_rng(77)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(6)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(3)
        e.upload_index_table(np.random.randint(0, N, size=(4, B)))
        e.graph_build(3)
        e.graph_run(3, 3)
        e.sync()
    st0 = algs[0].engine.read_stats()
    st0.pop("_device_ms")   # a timing, not a statistic
    assert all(np.isfinite(v) for v in st0.values())
    for other in algs[1:]:
        for name in ("online", "target", "adam_m", "adam_v"):
            assert torch.equal(getattr(algs[0].engine, name), getattr(other.engine, name)), name
        st1 = other.engine.read_stats()
        st1.pop("_device_ms")
        assert st0 == st1
    for k in env:
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("name", STEP_CASES)
def test_against_reference_golden(name):
    z, cfg, init = load_step_case(name)
    over = dict(auto_alpha=cfg["auto_alpha"], alpha=cfg["alpha"], delay_update=cfg["delay_update"])
    run_case("golden " + name, cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), int(z["cfg_batch"]),
             steps=int(z["cfg_steps"]), act_limit=float(z["cfg_act_limit"]), init=init, golden=z, **over)


def test_humanoid_b256_against_reference_digest():
    """BASELINE.json's own configuration against numbers the UNMODIFIED reference produced (tests/golden/
    step_humanoid_digest.npz, oracle/make_golden.py): tb_info, every 499th gradient / parameter / target element,
    per-tensor gradient norms -- same gates as the oracle comparisons above."""
    from dsac_v2_hip import DSAC_V2_HIP

    z, cfg, init, steps = humanoid_digest()
    O, A, hid, B = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), int(z["cfg_batch"])
    alg = DSAC_V2_HIP(**hip_kwargs(O, A, hid, B, act_limit=float(z["cfg_act_limit"]), strict_rng=True))
    alg.networks.load_state_dict(init)
    e, lay, stride = alg.engine, alg.engine.layout, int(z["cfg_stride"])
    rep = Report("humanoid 3x256 B=256 vs reference digest")
    n_tot = 2 * lay.n_q + lay.n_pi + 1
    idx = np.arange(0, n_tot, stride)
    bounds = np.cumsum([0, lay.n_q, lay.n_q, lay.n_pi, 1])
    names = ("q1", "q2", "policy", "log_alpha")
    seg_of = np.searchsorted(bounds, idx, side="right") - 1
    lrs = (cfg["lr_q"], cfg["lr_q"], cfg["lr_pi"], cfg["lr_alpha"])
    noise_b = AdamNoise([(names[k], int((seg_of == k).sum()), lrs[k]) for k in range(4)])
    crit = TB_KEYS.index("Loss/Critic loss-RL iter")
    keep_i = [i for i in range(len(TB_KEYS) - 1) if i != crit]
    for it, (data, noise) in enumerate(steps):
        e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
        e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
        e.compute_grads(it)
        e.sync()
        g = e.grads.cpu().numpy()[:n_tot]
        gs, gs_ref, gmax = g[idx], z["s%d/grad_s" % it], z["s%d/grad_max" % it]
        for k in range(3):
            m = seg_of == k
            assert m.any()
            rep.cmp("it%d grad.%s (sampled)" % (it, names[k]), gs[m], gs_ref[m], 1e-9 + 3e-5 * float(gmax[k]))
        # the stride misses the arena's last element: log_alpha's gradient is pinned on its own (the alpha gradient is
        # -mean(logp + target_entropy), dsac_v2.py:312-318 -- a mean of B terms of size ~10: 1e-4 absolute like tb_info)
        rep.cmp("it%d grad.log_alpha" % it, g[n_tot - 1:n_tot], z["s%d/grad_log_alpha" % it], 1e-4)
        # per-tensor L2 norms of the whole gradient (state_dict order inside each net)
        l2, off = [], 0
        for net_name, sd_prefix in (("q1", "q1."), ("q2", "q2."), ("policy", "policy.")):
            for k_, v in init.items():
                if k_.startswith(sd_prefix) and (k_.endswith(".weight") or k_.endswith(".bias")):
                    n = v.numel()
                    l2.append(float(np.linalg.norm(g[off:off + n].astype(np.float64))))
                    off += n
        rep.cmp("it%d per-tensor |grad|_2" % it, np.array(l2) / z["s%d/grad_l2" % it], np.ones(len(l2)), 3e-5)
        delayed = it % cfg["delay_update"] == 0
        noise_b.step(gs_ref, gs, ("q1", "q2") + (("policy", "log_alpha") if delayed else ()))
        e.apply_update(it)
        st = e.read_stats()
        tb_g = np.asarray(z["s%d/tb" % it], np.float64)
        rep.cmp("it%d tb vs reference" % it, [st[TB_KEYS[i]] for i in keep_i], tb_g[keep_i], 1e-4)
        rep.cmp("it%d critic loss vs reference" % it, [st[TB_KEYS[crit]]], [tb_g[crit]], 1e-6, 1e-5)
        p_hip, t_hip = e.online.cpu().numpy()[:n_tot], e.target.cpu().numpy()
        rep.cmp_params("it%d params vs reference (sampled)" % it, p_hip[idx], z["s%d/params_s" % it], noise_b.bound, 1e-6,
                       noise_b.lr_steps)
        rep.cmp("it%d log_alpha vs reference" % it, p_hip[n_tot - 1:n_tot], z["s%d/log_alpha" % it], 1e-6)
        n_t = t_hip.size
        idx_t = idx[idx < n_t]
        tau = cfg["tau"]
        rep.cmp_params("it%d targets vs reference (sampled)" % it, t_hip[idx_t], z["s%d/targets_s" % it],
                       tau * (it + 1) * noise_b.bound[:idx_t.size], 1e-7, tau * (it + 1) * noise_b.lr_steps)
        sums, off = [], 0
        for sd_prefix in ("q1.", "q2.", "policy."):
            for k_, v in init.items():
                if k_.startswith(sd_prefix) and (k_.endswith(".weight") or k_.endswith(".bias")):
                    n = v.numel()
                    sums.append(float(np.abs(p_hip[off:off + n].astype(np.float64)).sum()))
                    off += n
        rep.cmp("it%d per-tensor sum|param|" % it, np.array(sums) / z["s%d/param_abs_sums" % it], np.ones(len(sums)), 1e-5)
    rep.finish()


def test_local_update_surface_and_lazy_stats():
    alg, orc = make_pair(11, 3, (64, 64), 64)
    rng = np.random.default_rng(0)
    for it in range(4):
        data = synth_batch(rng, 64, 11, 3)
        torch.manual_seed(7 + it)
        noise = draw_noise(64, 3)
        torch.manual_seed(7 + it)
        tb = alg.local_update(data, it)  # strict_rng: draws the same 8 tensors from the global RNG
        ref = orc.local_update(data, noise, it)
        assert list(tb.keys()) == TB_KEYS
        for k in TB_KEYS[:-1]:
            assert abs(float(tb[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (it, k)
        assert torch.is_tensor(tb["DSAC2/mean_std1"])
    # a reference-style caller may log the PREVIOUS update's dict after issuing the next update: the statistics of the
    # last STATS_SLOTS-1 updates stay readable (snapshot ring), older ones raise
    torch.manual_seed(99)
    noise = draw_noise(64, 3)
    torch.manual_seed(99)
    held = alg.local_update(data, 4)
    ref = orc.local_update(data, noise, 4)
    very_old = held.__class__(alg, alg._serial, 0.0)
    for it in range(5, 8):
        torch.manual_seed(100 + it)
        alg.local_update(data, it)
    for k in TB_KEYS[:-1]:
        assert abs(float(held[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), k
    for it in range(8, 8 + alg.engine.STATS_SLOTS):
        alg.local_update(data, it)
    with pytest.raises(RuntimeError):
        very_old["Loss/Critic loss-RL iter"]


def test_adjustable_parameters_reach_the_engine():
    """dsac_v2.py:92-99: gamma / tau / alpha / auto_alpha / delay_update are re-read by the reference on every update;
    assigning them on the HIP algorithm must change the NEXT update the same way."""
    alg, orc = make_pair(11, 3, (64, 64), 64, seed=2)
    assert set(alg.adjustable_parameters) == {"gamma", "tau", "auto_alpha", "alpha", "delay_update"}
    rng = np.random.default_rng(3)
    changes = {1: ("gamma", 0.9), 2: ("tau", 0.05), 3: ("delay_update", 1), 4: ("auto_alpha", False), 5: ("alpha", 0.7)}
    for it in range(7):
        if it in changes:
            name, val = changes[it]
            setattr(alg, name, val)
            assert getattr(alg, name) == val
            orc.cfg[name] = val
        data = synth_batch(rng, 64, 11, 3)
        torch.manual_seed(40 + it)
        noise = draw_noise(64, 3)
        torch.manual_seed(40 + it)
        tb = alg.local_update(data, it)
        ref = orc.local_update(data, noise, it)
        for k in TB_KEYS[:-1]:
            assert abs(float(tb[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (it, k)
    alg.engine.sync()
    assert np.abs(alg.engine.target.cpu().numpy() - orc.flat_targets().numpy()).max() < 1e-5   # tau and delay_update took effect
    with pytest.raises(Exception):
        alg.delay_update = 0
    assert alg.delay_update == 1


def test_remote_flow_reports_fresh_mean_std_and_foreign_gradients_leave_it_alone():
    """get_remote_update_info's tb_info carries the mean_std the loss just used (dsac_v2.py:201-202), not the previous
    EMA; remote_update on a learner that did not compute the gradient never touches mean_std."""
    worker, orc = make_pair(11, 3, (64, 64), 64, seed=6)
    learner, _ = make_pair(11, 3, (64, 64), 64, seed=6)
    rng = np.random.default_rng(8)
    for it in range(3):
        data = synth_batch(rng, 64, 11, 3)
        torch.manual_seed(70 + it)
        noise = draw_noise(64, 3)
        torch.manual_seed(70 + it)
        tb, info = worker.get_remote_update_info(data, it)
        ref = orc.compute_gradient(data, noise)
        for k in ("DSAC2/mean_std1", "DSAC2/mean_std2"):
            assert abs(float(tb[k]) - float(ref[k])) <= 1e-5, (it, k)   # read BEFORE the update is applied
        learner.remote_update({k: ([g.clone() for g in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v))
                               for k, v in info.items()})
        worker.remote_update(info)
        orc.update(it)
    learner.engine.sync(); worker.engine.sync()
    assert torch.equal(learner.engine.online, worker.engine.online)
    assert learner.engine.get_state()["mean_std"] == [-1.0, -1.0]            # never initialised on the pure learner
    w = worker.engine.get_state()["mean_std"]
    assert abs(w[0] - float(orc.mean_std1)) < 1e-5 and abs(w[1] - float(orc.mean_std2)) < 1e-5


def test_sampling_ahead_restages_the_older_token():
    """two sample_batch() calls before local_update(): the engine stages one minibatch at a time, so the older token
    re-gathers its rows (same indices) instead of silently training on the newer ones."""
    from training.hip_replay_buffer import HipReplayBuffer

    O, A, B, N = 11, 3, 64, 512
    alg, _ = make_pair(O, A, (64, 64), B, seed=5)
    kw = hip_kwargs(O, A, (64, 64), B, buffer_max_size=N)
    buf = HipReplayBuffer(**kw)
    assert buf.engine is alg.engine
    rng = np.random.default_rng(2)
    buf.add_batch([(rng.standard_normal(O).astype(np.float32), {}, rng.uniform(-.4, .4, A).astype(np.float32), float(i),
                    rng.standard_normal(O).astype(np.float32), False, 0.0, {}) for i in range(N)])
    np.random.seed(3)
    first, second = buf.sample_batch(B), buf.sample_batch(B)
    want_first = first.idxs.astype(np.float32)      # the reward of row i is i
    alg.local_update(first, 0)
    alg.engine.sync()
    assert np.array_equal(alg.engine.read_batch()["rew"], want_first)
    assert np.array_equal(second["rew"].numpy(), second.idxs.astype(np.float32))
    assert np.array_equal(first["rew"].numpy(), want_first)
    # rows replaced after sampling: the reference's batch is a copy taken at sample time; a token that is no
    # longer the staged minibatch re-gathers by index and must refuse once add_batch has overwritten its rows
    third, fourth = buf.sample_batch(B), buf.sample_batch(B)
    buf.add_batch([(rng.standard_normal(O).astype(np.float32), {}, rng.uniform(-.4, .4, A).astype(np.float32), -1.0,
                    rng.standard_normal(O).astype(np.float32), False, 0.0, {}) for _ in range(N)])
    with pytest.raises(RuntimeError, match="overwritten"):
        alg.local_update(third, 1)
    alg.local_update(fourth, 1)      # still the staged minibatch: the staging area IS the copy taken at sample time
    alg.engine.sync()
    assert np.array_equal(alg.engine.read_batch()["rew"], fourth.idxs.astype(np.float32))


def test_remote_update_seam_equals_local_update():
    a1, _ = make_pair(11, 3, (64, 64), 64, seed=3)
    a2, _ = make_pair(11, 3, (64, 64), 64, seed=3)
    rng = np.random.default_rng(1)
    for it in range(3):
        data = synth_batch(rng, 64, 11, 3)
        torch.manual_seed(50 + it)
        a1.local_update(data, it)
        torch.manual_seed(50 + it)
        _, info = a2.get_remote_update_info(data, it)
        assert len(info["q1_grad"]) == 6 and info["q1_grad"][0].shape == (64, 14)
        a2.remote_update(info)
    a1.engine.sync(); a2.engine.sync()
    assert torch.equal(a1.engine.online, a2.engine.online)
    assert torch.equal(a1.engine.target, a2.engine.target)


def test_replay_ring_and_gather_bit_exact():
    from plugin import create_buffer

    O, A, N, B = 13, 2, 50, 16
    alg, _ = make_pair(O, A, (32,), B)
    buf = create_buffer(**hip_kwargs(O, A, (32,), B, buffer_max_size=N))
    assert buf.engine is alg.engine
    orc = ReplayOracle(O, A, N)
    rng = np.random.default_rng(3)
    samples = [(rng.standard_normal(O).astype(np.float32), {}, rng.uniform(-1, 1, A).astype(np.float32),
                float(rng.standard_normal()), rng.standard_normal(O).astype(np.float32), bool(rng.random() < 0.2),
                np.float32(rng.standard_normal()), {}) for _ in range(123)]
    for lo, hi in ((0, 30), (30, 31), (31, 73), (73, 123)):  # wraps the ring twice
        buf.add_batch(samples[lo:hi])
        orc.add_batch(samples[lo:hi])
        assert (buf.size, buf.ptr) == (orc.size, orc.ptr)
        np.random.seed(11 + lo)
        got = buf.sample_batch(B)
        np.random.seed(11 + lo)
        want = orc.sample_batch(B)
        for k in ("obs", "obs2", "act", "rew", "done", "logp"):
            assert torch.equal(got[k], want[k]), k
    # golden vectors produced by the reference ReplayBuffer itself
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "replay.npz"))
    O, A, N = [int(v) for v in z["cfg"]]
    alg2, _ = make_pair(O, A, (32,), 16)
    buf2 = create_buffer(**hip_kwargs(O, A, (32,), 16, buffer_max_size=N))
    s = []
    for i in range(73):
        r, d, l = z["in/rdl%d" % i]
        s.append((z["in/obs%d" % i], {}, z["in/act%d" % i], float(r), z["in/obs2_%d" % i], bool(d), np.float32(l), {}))
    buf2.add_batch(s[:30])
    np.random.seed(11)
    b = buf2.sample_batch(16)
    for k in ("obs", "obs2", "act", "rew", "done", "logp"):
        np.testing.assert_array_equal(b[k].numpy(), z["b30/" + k])
    buf2.add_batch(s[30:])
    assert (buf2.size, buf2.ptr) == (int(z["size_73"]), int(z["ptr_73"]))
    b = buf2.sample_batch(16)
    for k in ("obs", "obs2", "act", "rew", "done", "logp"):
        np.testing.assert_array_equal(b[k].numpy(), z["b73/" + k])


def test_buffer_fast_path_equals_host_path():
    """HipReplayBuffer token (minibatch stays in HBM) == reference-style dict of CPU tensors."""
    from plugin import create_buffer

    O, A, B = 11, 3, 64
    a1, _ = make_pair(O, A, (64, 64), B, seed=2)
    buf = create_buffer(**hip_kwargs(O, A, (64, 64), B, buffer_max_size=500))
    rng = np.random.default_rng(9)
    n = 300
    buf.engine.buffer_add(rng.standard_normal((n, O), dtype=np.float32), rng.uniform(-.4, .4, (n, A)).astype(np.float32),
                          rng.standard_normal(n, dtype=np.float32), rng.standard_normal((n, O), dtype=np.float32),
                          (rng.random(n) < 0.1).astype(np.float32))
    a2, _ = make_pair(O, A, (64, 64), B, seed=2)
    for it in range(3):
        np.random.seed(it)
        tok = buf.sample_batch(B)
        host = {k: v.clone() for k, v in tok.items()}
        torch.manual_seed(it)
        a1.local_update(tok, it)
        torch.manual_seed(it)
        a2.local_update(host, it)
    a1.engine.sync(); a2.engine.sync()
    assert torch.equal(a1.engine.online, a2.engine.online)


@pytest.mark.parametrize("O,A,hid,B,per_graph,total", [
    (17, 4, (64, 64), 64, 2, 8),
    (17, 4, (64, 64), 64, 3, 6),          # odd updates per graph: both orders of the two batch sets
    (11, 3, (96, 40), 50, 4, 8),          # ragged widths / batch: edge tiles, a partial last gather block
    (5, 1, (33,), 7, 1, 4),               # one update per graph: nothing rides, only the bookkeeping block
    (376, 17, (256, 256, 256), 256, 8, 16),
    (17, 4, (64, 64), 64, 7, 14),
])
def test_graph_replay_equals_eager_steps(O, A, hid, B, per_graph, total):
    """Graph replays (one gather per graph; each update's loss launch stages the NEXT update's minibatch into the other
    batch set and does the bookkeeping; the first-layer weight tiles keep the padded copies fresh) == eager updates
    (own gather + repack per update), bit for bit -- parameters, targets, optimiser state, and the staged minibatch."""
    N = 4096
    algs = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hid, B, seed=4)
        e = alg.engine
        e.set_device_rng(12345)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        idx = np.random.randint(0, N, size=(8, B))
        e.upload_index_table(idx)
        if mode == "graph":
            e.graph_build(per_graph)
            e.graph_run(0, total)
        else:
            ms = e.time_steps(0, total, use_graph=False)
            assert ms > 0
        e.sync()
        algs.append(alg)
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(algs[0].engine, name), getattr(algs[1].engine, name)), name
    st = algs[1].engine.get_state()
    assert st == algs[0].engine.get_state()
    assert st["adam_steps"] == [total, (total + 1) // 2, (total + 1) // 2]
    assert torch.isfinite(algs[1].engine.online).all()
    # the minibatch of the LAST update is the one left staged, in both flows
    b0, b1 = algs[0].engine.read_batch(with_logp=False), algs[1].engine.read_batch(with_logp=False)
    for k in ("obs", "act", "rew", "obs2", "done"):
        assert np.array_equal(b0[k], b1[k]), k


PIPE_BUFFERS = ("logits_pi", "logp_new", "logp2", "qout_t0", "qout_t1", "qout_p0", "qout_c1", "XP", "X2", "eps_new", "z5",
                "part_heads", "part_loss", "H.pi.0", "G.pi.1", "dZ.pi.0", "d_new_act")


@pytest.mark.parametrize("O,A,hid,B,D,per_graph,first,total,env", [
    (16, 4, (64, 64), 64, 2, 4, 1, 8, {}),             # starts on an odd iteration: every update is half of a pair
    (16, 4, (64, 64), 64, 2, 4, 0, 8, {}),             # starts on an even one: single, pairs, single
    (16, 4, (64, 64), 64, 2, 3, 1, 9, {}),             # odd graph length: replays alternate between the two phase graphs
    (16, 4, (64, 64), 64, 2, 5, 2, 10, {"DSACT_PIPE_QT": "1"}),           # q_target of the next minibatch precomputed too
    (24, 6, (128, 128, 128), 32, 3, 6, 1, 12, {}),     # delay_update 3: a (has, makes) launch in the middle of every window
    (24, 6, (128, 128, 128), 32, 3, 4, 0, 12, {"DSACT_PIPE_QT": "1", "DSACT_PIPE_RG_SIDE": "1"}),
    (16, 4, (64, 64), 64, 1, 4, 0, 8, {}),             # delay_update 1: nothing to share, the plain graph
    (16, 4, (64, 64), 16, 2, 2, 1, 6, {"DSACT_PIPE_RG_NEXT": "1"}),       # 4 slices per unit
    (16, 4, (64, 64), 64, 2, 4, 1, 8, {"DSACT_NO_PIPE_DEFER": "1"}),      # the discarded policy backward stays in its own update
    (16, 4, (64, 64), 64, 2, 4, 1, 8, {"DSACT_NO_BQT_MERGE": "1"}),       # critics' backward and their tiles as two launches (round 4's form)
    (376, 17, (256, 256, 256), 256, 2, 8, 1, 16, {"DSACT_NO_BQT_MERGE": "1"}),
    (16, 4, (64, 64), 64, 2, 4, 0, 8, {"DSACT_NO_BQP_MERGE": "1"}),       # policy-moving updates: critics' and policy backward as two launches
    (24, 6, (128, 128, 128), 32, 3, 6, 2, 12, {"DSACT_NO_BQP_MERGE": "1"}),
    (32, 8, (256, 256), 48, 2, 5, 1, 10, {"DSACT_PIPE_QT": "1", "DSACT_PIPE_QP_SPLIT": "1"}),
    (376, 17, (256, 256, 256), 256, 2, 8, 1, 16, {}),  # the BASELINE.json shape
    (376, 17, (256, 256, 256), 256, 2, 6, 4, 12, {"DSACT_PIPE_MAP": "FT.pit=23:1;FT.q1t=01;TF.q1c=0123:2"}),
])
def test_pipelined_graph_equals_eager_steps(O, A, hid, B, D, per_graph, first, total, env, monkeypatch):
    """Delayed-update-aware pipelined graph (k_chain_fwdp: the forward launch of an update that leaves the policy alone
    also runs the policy units of the NEXT minibatch; the gather rides two updates ahead) == eager updates, bit for bit:
    parameters, targets, optimiser state, step state, statistics, the staged minibatch and the intermediates of the last
    update -- for graphs that start on either parity, odd graph lengths, delay_update 1 / 2 / 3, and placement variants."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N = 2048
    algs, stats, bufs = [], [], []
    for mode in ("eager", "graph", "sequence"):
        alg, _ = make_pair(O, A, hid, B, seed=4, delay_update=D)
        e = alg.engine
        assert e.chain_active
        e.set_device_rng(777)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(7, B)))
        if mode == "graph":
            e.graph_build(per_graph)
            assert e.debug_get("pipe_graph") == (1.0 if D >= 2 else 0.0)
            e.graph_run(first, total)
        elif mode == "sequence":
            # the same launch sequence issued eagerly with an event pair per dispatch (what bench.py profiles)
            names = [n for n, _, _ in e.profile_steps(first, total)]
            if D >= 2:
                assert "chain_fwd+next" in names and "chain_fwd_q" in names, names
                assert ("chain_bwd_qt" in names) == ("DSACT_NO_BQT_MERGE" not in env and "DSACT_NO_PIPE_DEFER" not in env), names
                assert ("chain_bwd_qpt" in names) == ("DSACT_NO_BQT_MERGE" not in env and "DSACT_NO_BQP_MERGE" not in env), names
        else:
            assert e.time_steps(first, total, use_graph=False) > 0
        e.sync()
        algs.append(alg)
        st = e.read_stats()
        stats.append({k: v for k, v in st.items() if not k.startswith("_device")})
        bufs.append({n: e.debug_read(n) for n in PIPE_BUFFERS})
    for other in (1, 2):
        for name in ("online", "target", "adam_m", "adam_v"):
            assert torch.equal(getattr(algs[0].engine, name), getattr(algs[other].engine, name)), (other, name)
        assert algs[0].engine.get_state() == algs[other].engine.get_state()
        assert torch.isfinite(algs[other].engine.online).all()
        for k in stats[0]:
            assert stats[0][k] == stats[other][k] or (np.isnan(stats[0][k]) and np.isnan(stats[other][k])), (other, k, stats[0][k], stats[other][k])
        for n in PIPE_BUFFERS:
            assert np.array_equal(bufs[0][n], bufs[other][n]), (other, n)
        b0, b1 = algs[0].engine.read_batch(with_logp=False), algs[other].engine.read_batch(with_logp=False)
        for k in ("obs", "act", "rew", "obs2", "done"):
            assert np.array_equal(b0[k], b1[k]), (other, k)
    # a second run on the same handle continues identically (the phase graphs are cached; the other phase is picked)
    algs[0].engine.time_steps(first + total, per_graph, use_graph=False)
    algs[1].engine.graph_run(first + total, per_graph)
    algs[0].engine.sync(); algs[1].engine.sync()
    assert torch.equal(algs[0].engine.online, algs[1].engine.online)
    assert torch.equal(algs[0].engine.target, algs[1].engine.target)


@pytest.mark.parametrize("O,A,hid,B", [(24, 6, (64, 64), 64), (376, 17, (256, 256, 256), 256), (11, 3, (96, 40), 50)])
def test_gauss_distribution(O, A, hid, B):
    """policy_act_distribution = "GaussDistribution" (utils/act_distribution_cls.py:82-115, a kwarg of rows a10 / a11 the HIP
    path refused until round 5): action = mean + std * eps without squashing, log-prob of the plain diagonal Gaussian -- every
    intermediate, gradient, statistic and parameter against the oracle (pinned bit-exact to the live reference with this
    kwarg, tests/test_oracle_vs_reference.py) on the chain path, the BASELINE shape and the tile path."""
    run_case("GaussDistribution O=%d A=%d hid=%s B=%d" % (O, A, hid, B), O, A, hid, B, steps=3, policy_act_distribution="GaussDistribution")


def test_gauss_distribution_sampling_step():
    """dsact_act_sample with the plain Gaussian == GaussDistribution.sample() on the same logits and generator state, and the
    container hands the sampler / evaluator that class (mode() = clamp(mean, limits))."""
    from dsac_v2_hip import GaussDistribution

    O, A, hid, B, lim = 24, 6, (64, 64), 64, 0.4
    alg, _ = make_pair(O, A, hid, B, act_limit=lim, seed=3, policy_act_distribution="GaussDistribution")
    e = alg.engine
    rng = np.random.default_rng(2)
    for i in range(10):
        obs = rng.standard_normal(O).astype(np.float32)
        torch.manual_seed(i)
        eps = torch.randn(1, A)
        action, logp = e.act_sample(obs, eps.numpy())
        logits = torch.from_numpy(e.policy_forward(obs[None]))
        dist = alg.networks.create_action_distributions(logits)
        assert isinstance(dist, GaussDistribution)
        torch.manual_seed(i)
        a_ref, lp_ref = dist.sample()
        np.testing.assert_allclose(action, a_ref[0].numpy(), atol=2e-6, rtol=0)
        assert abs(float(logp[0]) - float(lp_ref[0])) <= 2e-4
        assert torch.equal(dist.mode(), torch.clamp(dist.mean, -lim, lim))


def test_device_rng_is_standard_normal():
    alg, _ = make_pair(8, 8, (32,), 1024)
    e = alg.engine
    e.set_device_rng(99)
    data = synth_batch(np.random.default_rng(0), 1024, 8, 8)
    e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
    e.step(0)
    a = np.concatenate([e.debug_read("eps_new"), e.debug_read("eps_2"), e.debug_read("z5"), e.debug_read("z6")])
    assert abs(a.mean()) < 0.03 and abs(a.std() - 1.0) < 0.03
    e.step(1)
    b = e.debug_read("eps_new")
    assert not np.array_equal(a[: b.size], b)


def test_policy_forward_and_checkpoint_roundtrip(tmp_path):
    from dsac_v2_hip import ApproxContainer

    O, A, hid, B = 376, 17, (256, 256, 256), 256
    alg, orc = make_pair(O, A, hid, B)
    obs = torch.randn(5, O)
    lg = alg.networks.policy(obs)
    from oracle.dsact_oracle import policy_forward
    want = policy_forward(obs, [p.detach() for p in orc.p["policy"]], orc.cfg)
    assert torch.allclose(lg, want, atol=2e-5, rtol=1e-5)
    dist = alg.networks.create_action_distributions(lg)
    act, logp = dist.sample()
    assert act.shape == (5, A) and logp.shape == (5,) and act.abs().max() <= 0.4 + 1e-6
    # state_dict format == reference checkpoints
    import json
    lay = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_layout.json")))
    sd = {k: v.clone() for k, v in alg.networks.state_dict().items()}  # state_dict() aliases the arenas
    assert [[k, list(v.shape)] for k, v in sd.items()] == lay["humanoid_l3"]
    p = str(tmp_path / "apprfunc_0.pkl")
    torch.save(sd, p)
    data = synth_batch(np.random.default_rng(0), B, O, A)
    torch.manual_seed(0)
    alg.local_update(data, 0)
    changed = alg.networks.state_dict()
    assert not torch.equal(changed["q1.q.0.weight"], sd["q1.q.0.weight"])
    alg.networks.load_state_dict(torch.load(p))
    back = alg.networks.state_dict()
    for k in sd:
        assert torch.equal(back[k].cpu(), sd[k].cpu()), k
    # a stand-alone CPU container (sampler / evaluator / PolicyRunner) loads the same file
    cpu = ApproxContainer(**hip_kwargs(O, A, hid, B))
    cpu.load_state_dict(torch.load(p, map_location="cpu"))
    assert torch.allclose(cpu.policy(obs), lg, atol=2e-5, rtol=1e-5)


def test_single_launch_acting_forward(monkeypatch):
    """dsact_act.h: the sampler's batch-1 `networks.policy(obs)` as one launch (observation in the kernel arguments,
    logits through mapped host memory) == the copy + tile-stage path, follows the live weights through updates, and
    survives thousands of calls (monotone arrival counters)."""
    O, A, hid, B = 376, 17, (256, 256, 256), 256
    alg, orc = make_pair(O, A, hid, B, seed=51)
    e = alg.engine
    from oracle.dsact_oracle import policy_forward
    rng = np.random.default_rng(3)
    obs = rng.standard_normal((64, O)).astype(np.float32)
    slow = e.policy_forward(obs)                       # n = 64: the batched path
    fast = np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(64)])
    assert fast.shape == slow.shape == (64, 2 * A)
    np.testing.assert_allclose(fast, slow, atol=2e-6, rtol=1e-6)
    want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
    np.testing.assert_allclose(fast, want, atol=2e-5, rtol=1e-5)
    # the forward reads the parameter arena itself: after an update (issued asynchronously on the same stream) the
    # next call sees the new policy
    data = synth_batch(rng, B, O, A)
    torch.manual_seed(1)
    alg.local_update(data, 0)
    after = e.policy_forward(obs[:1])
    assert np.abs(after - fast[:1]).max() > 1e-6
    sd = [p.detach().cpu() for p in alg.networks.policy.policy.parameters()]
    x = torch.as_tensor(obs[:1])
    for k in range(0, len(sd) - 2, 2):
        x = torch.nn.functional.gelu(x @ sd[k].T + sd[k + 1])
    out = x @ sd[-2].T + sd[-1]
    ref = torch.cat([out[:, :A], torch.clamp(out[:, A:], -20.0, 0.5).exp()], dim=-1).numpy()
    np.testing.assert_allclose(after, ref, atol=2e-5, rtol=1e-5)
    for i in range(3000):                                # counters, mapped completion word, no leaks
        got = e.policy_forward(obs[i % 64:i % 64 + 1])
    np.testing.assert_allclose(got, e.policy_forward(obs[(2999 % 64):(2999 % 64) + 1]), atol=0, rtol=0)
    assert e.debug_get("handoff_failures") == 0.0
    # narrow / ragged nets take the same path
    alg2, orc2 = make_pair(5, 1, (33,), 7, act_limit=2.0, seed=52)
    o2 = rng.standard_normal((3, 5)).astype(np.float32)
    f2 = np.concatenate([alg2.engine.policy_forward(o2[i:i + 1]) for i in range(3)])
    w2 = policy_forward(torch.as_tensor(o2), [p.detach() for p in orc2.p["policy"]], orc2.cfg).numpy()
    np.testing.assert_allclose(f2, w2, atol=2e-5, rtol=1e-5)


def test_act_sample_is_the_reference_sampling_step():
    """dsact_act_sample (acting forward + TanhGaussDistribution.sample() in the output layer's epilogue, dsact_act.h) against
    the reference's own sequence on the same logits and the same N(0,1) draw (training/off_sampler.py:46-54,
    utils/act_distribution_cls.py:32-42), and the sampler's fast path (one call per environment step, one torch.randn per
    step, packed transitions) against its general path from the same seeds: same generator consumption, same transitions."""
    from dsac_v2_hip import TanhGaussDistribution
    from training.hip_sampler import HipOffSampler

    for O, A, hid, B, lim in ((376, 17, (256, 256, 256), 256, 0.4), (5, 1, (33,), 7, 2.0), (24, 6, (128, 128), 64, 1.0)):
        alg, _ = make_pair(O, A, hid, B, act_limit=lim, seed=61)
        e = alg.engine
        rng = np.random.default_rng(2)
        for i in range(20):
            obs = (3.0 * rng.standard_normal(O)).astype(np.float32)      # large inputs: some actions near their limits
            torch.manual_seed(i)
            eps = torch.randn(1, A)
            action, logp = e.act_sample(obs, eps.numpy())
            logits = torch.from_numpy(e.policy_forward(obs[None]))
            dist = TanhGaussDistribution(logits)
            dist.act_high_lim, dist.act_low_lim = alg.networks.policy.act_high_lim.cpu(), alg.networks.policy.act_low_lim.cpu()
            torch.manual_seed(i)
            a_ref, lp_ref = dist.sample()                                  # draws the same eps from the same generator state
            np.testing.assert_allclose(action, a_ref[0].numpy(), atol=2e-6 * lim, rtol=0)
            t2 = (np.asarray(a_ref[0], np.float64) / lim) ** 2             # saturation budget as in compare_intermediates
            tol = 2e-4 + float((2.4e-7 / (1.0 + 1e-6 - np.minimum(t2, 1.0))).sum())
            assert abs(float(logp[0]) - float(lp_ref[0])) <= tol, (i, float(logp[0]), float(lp_ref[0]), tol)

    class Env:     # deterministic toy dynamics with a time limit
        class _S:
            low, high = np.full(4, -0.3, np.float32), np.full(4, 0.3, np.float32)
        action_space = _S()

        def __init__(self):
            self.t, self.s = 0, np.zeros(16, np.float32)

        def reset(self):
            self.t, self.s = 0, np.linspace(-1, 1, 16).astype(np.float32)
            return self.s.copy(), {}

        def step(self, a):
            self.t += 1
            self.s = (0.9 * self.s + 0.1 * np.resize(a, 16)).astype(np.float32)
            trunc = self.t >= 7
            return self.s.copy(), float(self.s.sum()), bool(abs(self.s[0]) > 5), {"TimeLimit.truncated": trunc}

    alg, _ = make_pair(16, 4, (64, 64), 32, act_limit=0.4, seed=62)
    outs = []
    for fast in (True, False):
        smp = HipOffSampler(env=Env(), networks=alg.networks, sample_batch_size=25, action_type="continu")
        if not fast:
            smp._fast_engine = lambda: None
        torch.manual_seed(9)
        batch, _ = smp.sample()
        outs.append((batch, torch.randn(2)))
        assert (getattr(batch, "packed", None) is not None) == fast
    (b0, r0), (b1, r1) = outs
    assert torch.equal(r0, r1)                                             # the generator was consumed identically
    assert len(b0) == len(b1) == 25
    for s0, s1 in zip(b0, b1):
        np.testing.assert_allclose(s0[0], s1[0], atol=1e-5)
        np.testing.assert_allclose(s0[2], s1[2], atol=2e-6)
        assert abs(s0[3] - s1[3]) < 1e-4 and bool(s0[5]) == bool(s1[5]) and abs(float(s0[6]) - float(s1[6])) < 5e-4
        np.testing.assert_allclose(s0[4], s1[4], atol=1e-5)
        assert s0[7]["TimeLimit.truncated"] == s1[7]["TimeLimit.truncated"]
    obs_p, act_p, rew_p, obs2_p, done_p, logp_p = b0.packed
    assert obs_p.shape == (25, 16) and act_p.dtype == np.float32
    np.testing.assert_array_equal(obs_p[3], np.asarray(b0[3][0]).reshape(-1))
    np.testing.assert_array_equal(act_p[3], b0[3][2])


def _replay_pair(O, A, hid, B, N, seed):
    alg, _ = make_pair(O, A, hid, B, seed=seed)
    e = alg.engine
    e.set_device_rng(4242)
    e.buffer_create(N)
    g = torch.Generator(device="cuda").manual_seed(7)
    e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                         torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                         (torch.rand(N, device="cuda", generator=g) < .05).float())
    np.random.seed(3)
    e.upload_index_table(np.random.randint(0, N, size=(8, B)))
    return alg


def test_fused_optimizer_equals_split_update():
    """dsact_step (Adam/Polyak fused into the weight-gradient tiles) == compute_grads + apply_update (k_adam)."""
    O, A, hid, B = 23, 5, (64, 96, 64), 64
    a1, _ = make_pair(O, A, hid, B, seed=9)
    a2, _ = make_pair(O, A, hid, B, seed=9)
    rng = np.random.default_rng(2)
    for it in range(5):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(77 + it)
        noise = draw_noise(B, A)
        for a in (a1, a2):
            a.engine.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            a.engine.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
        a1.engine.step(it)
        a2.engine.compute_grads(it)
        a2.engine.apply_update(it)
    a1.engine.sync(); a2.engine.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a1.engine, name), getattr(a2.engine, name)), name
    assert a1.engine.get_state() == a2.engine.get_state()


def test_split_k_step_equals_split_k_halves():
    """batch 512 (two 256-sample chunks): dsact_step (k_adam sums the chunk partials itself) ==
    compute_grads (k_sum_parts -> gradient arena) + apply_update, bit for bit."""
    O, A, hid, B = 23, 5, (64, 96, 64), 512
    a1, _ = make_pair(O, A, hid, B, seed=9)
    a2, _ = make_pair(O, A, hid, B, seed=9)
    rng = np.random.default_rng(2)
    for it in range(4):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(77 + it)
        noise = draw_noise(B, A)
        for a in (a1, a2):
            a.engine.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            a.engine.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
        a1.engine.step(it)
        a2.engine.compute_grads(it)
        a2.engine.apply_update(it)
    a1.engine.sync(); a2.engine.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a1.engine, name), getattr(a2.engine, name)), name
    assert a1.engine.get_state() == a2.engine.get_state()


def test_skip_discarded_actor_backward_keeps_trajectory():
    """DSACT_F_SKIP_ACTOR_ON_OFF_ITERS drops work whose result the reference throws away
    (dsac_v2.py:174-186 vs :324): parameters, targets and optimiser state must not change by a bit."""
    from dsact._ffi import F_SKIP_ACTOR_ON_OFF_ITERS
    outs = []
    for flags in (0, F_SKIP_ACTOR_ON_OFF_ITERS):
        alg = _replay_pair(17, 4, (64, 64), 64, 4096, seed=4)
        e = alg.engine
        e.graph_build(2, flags)
        e.graph_run(0, 8)
        e.sync()
        outs.append((e.online.clone(), e.target.clone(), e.adam_m.clone(), e.adam_v.clone(), e.get_state(), e.read_stats()))
    for x, y in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(x, y)
    assert outs[0][4] == outs[1][4]
    for k in outs[0][5]:
        assert outs[0][5][k] == outs[1][5][k], k


def test_data_parallel_halves_equal_graph_replay():
    """dsact_dp_enqueue_grads + (all-reduce) + dsact_dp_enqueue_apply through the DataParallelUpdater
    (world size 1 here: gpurun exposes one GPU) == the fused graph replay, bit for bit."""
    import torch.distributed as dist
    from dsact.dp import DataParallelUpdater

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        a1 = _replay_pair(17, 4, (64, 64), 64, 4096, seed=4)
        a1.engine.graph_build(2)
        a1.engine.graph_run(0, 6)
        a1.engine.sync()
        a2 = _replay_pair(17, 4, (64, 64), 64, 4096, seed=4)
        e = a2.engine
        dp = DataParallelUpdater(e, broadcast_tensors=(e.online, e.target, e.adam_m, e.adam_v))
        dp.force_collective = True   # one rank: still issue the all-reduce -- it runs on the collective's own stream
        e.dp_begin(0)                # and must be ordered against the engine's kernels both ways
        for _ in range(6):
            dp.step()
        torch.cuda.synchronize()
        for name in ("online", "target", "adam_m", "adam_v"):
            assert torch.equal(getattr(a1.engine, name), getattr(e, name)), name
        assert a1.engine.get_state() == e.get_state()
        # overlapped variant: critic half -> async all-reduce(q1|q2) -> actor half -> all-reduce(rest) -> apply.
        # Repeated: an ordering bug between the engine's stream and the collective's shows up as a flaky mismatch
        # (it did: 8 of 12 runs before the collectives were issued on engine.torch_stream)
        for rep_i in range(6):
            a3 = _replay_pair(17, 4, (64, 64), 64, 4096, seed=4)
            e3 = a3.engine
            dp3 = DataParallelUpdater(e3, broadcast_tensors=(e3.online, e3.target, e3.adam_m, e3.adam_v), overlap=True)
            assert dp3.overlap
            e3.dp_begin(0)
            for _ in range(6):
                dp3.step()
            torch.cuda.synchronize()
            for name in ("online", "target", "adam_m", "adam_v"):
                assert torch.equal(getattr(a1.engine, name), getattr(e3, name)), (name, rep_i)
            assert a1.engine.get_state() == e3.get_state()
    finally:
        if created:
            dist.destroy_process_group()


def test_native_rccl_graph_captured_data_parallel_update():
    """dsact_comm_init (the library's own RCCL communicator) + dsact_graph_build(DSACT_F_DATA_PARALLEL): gather ->
    gradients -> ncclAllReduce -> Adam/Polyak captured in ONE hipGraph (BASELINE.json configs[4]). World size 1 here
    (gpurun exposes one GPU): the average over one rank is the identity, so the result must equal the fused
    single-GPU graph replay bit for bit -- with the collective really in the graph; also the eager native path."""
    import torch.distributed as dist
    from dsact.dp import DataParallelUpdater

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        for shape in ((17, 4, (64, 64), 64), (376, 17, (256, 256, 256), 256)):
            O, A, hid, B = shape
            a1 = _replay_pair(O, A, hid, B, 4096, seed=4)
            a1.engine.graph_build(2)
            a1.engine.graph_run(0, 6)
            a1.engine.sync()
            # graph-captured data-parallel update
            a2 = _replay_pair(O, A, hid, B, 4096, seed=4)
            e2 = a2.engine
            dp2 = DataParallelUpdater(e2, broadcast_tensors=(e2.online, e2.target, e2.adam_m, e2.adam_v), native=True)
            dp2.build_graph(3)
            dp2.run_graph(0, 6)
            e2.sync()
            torch.cuda.synchronize()
            for name in ("online", "target", "adam_m", "adam_v"):
                assert torch.equal(getattr(a1.engine, name), getattr(e2, name)), (shape, "graph", name)
            assert a1.engine.get_state() == e2.get_state()
            # eager halves with the native all-reduce in the seam
            a3 = _replay_pair(O, A, hid, B, 4096, seed=4)
            e3 = a3.engine
            dp3 = DataParallelUpdater(e3, broadcast_tensors=(e3.online, e3.target, e3.adam_m, e3.adam_v), native=True)
            dp3.force_collective = True
            e3.dp_begin(0)
            for _ in range(6):
                dp3.step()
            e3.sync()
            torch.cuda.synchronize()
            for name in ("online", "target", "adam_m", "adam_v"):
                assert torch.equal(getattr(a1.engine, name), getattr(e3, name)), (shape, "eager", name)
            e2.comm_destroy(); e3.comm_destroy()
    finally:
        if created:
            dist.destroy_process_group()


def test_strict_data_parallel_shards_equal_global_batch():
    """SURVEY.md section 8e, strict mode: two replicas on the halves of one minibatch, with the 2-float
    pre-loss exchange of {sum std1, sum std2} (emulated in-process: gpurun exposes one GPU), average to the
    gradient of ONE engine on the whole minibatch -- sentinel step (mean_std = -1) included -- and agree on
    the mean_std EMA. The one-collective ("fast") mode differs on that step by the local-vs-global mean."""
    O, A, hid, B, N = 17, 4, (64, 64), 64, 512
    from dsac_v2_hip import DSAC_V2_HIP

    def build(batch):
        torch.manual_seed(11)
        alg = DSAC_V2_HIP(**hip_kwargs(O, A, hid, batch, strict_rng=True, global_batch=B))
        e = alg.engine
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(7)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        return alg

    full, h0, h1 = build(B), build(B // 2), build(B // 2)
    np.random.seed(3)
    idx = np.random.randint(0, N, size=(1, B))
    torch.manual_seed(5)
    noise = draw_noise(B, A)
    n = {k: noise[k].numpy() for k in ("eps_new", "eps_2", "z5", "z6")}
    full.engine.upload_index_table(idx)
    full.engine.set_noise(n["eps_new"], n["eps_2"], n["z5"], n["z6"])
    halves = (h0.engine, h1.engine)
    for r, e in enumerate(halves):
        lo, hi = r * B // 2, (r + 1) * B // 2
        e.upload_index_table(idx[:, lo:hi])
        e.set_noise(n["eps_new"][lo:hi], n["eps_2"][lo:hi], n["z5"][lo:hi], n["z6"][lo:hi])
        e.dp_set_strict(True)
    fe = full.engine
    fe.dp_begin(0)
    fe.dp_grads()
    fe.sync()
    for e in halves:
        e.dp_begin(0)
        e.dp_forward()
        e.sync()
    total = halves[0].std_sums + halves[1].std_sums     # the all-reduce(SUM) of the real job
    for e in halves:
        e.std_sums.copy_(total)
    torch.cuda.synchronize()
    for e in halves:
        e.dp_backward()
        e.sync()
    g_avg = (halves[0].grads + halves[1].grads) / 2
    nn_ = fe.layout.n_online
    rep = Report("strict data-parallel halves vs global batch")
    rep.cmp("grad (avg of shards)", g_avg[:nn_].cpu().numpy(), fe.grads[:nn_].cpu(), 1e-9, 5e-6)
    rep.cmp("mean_std (tail of the arena)", g_avg[nn_:].cpu().numpy(), fe.grads[nn_:].cpu(), 1e-7)
    assert torch.equal(halves[0].grads[nn_:], halves[1].grads[nn_:])
    # fast mode on the same shards: not equal on the sentinel step (documents why strict exists)
    f0 = build(B // 2).engine
    f0.upload_index_table(idx[:, :B // 2])
    f0.set_noise(n["eps_new"][:B // 2], n["eps_2"][:B // 2], n["z5"][:B // 2], n["z6"][:B // 2])
    f0.dp_begin(0)
    f0.dp_grads()
    f0.sync()
    assert not torch.equal(f0.grads[nn_:], halves[0].grads[nn_:])
    rep.finish()


def test_error_paths_are_loud():
    from dsact._ffi import DsactError
    from dsact.engine import DsactEngine

    e = DsactEngine(7, 2, (32,), 8)
    with pytest.raises(DsactError):       # arenas bound, but no action limits / minibatch yet
        e.step(0)
    e.set_action_limits(np.ones(2, np.float32), -np.ones(2, np.float32))
    with pytest.raises(DsactError):       # no minibatch staged
        e.step(0)
    with pytest.raises(DsactError):       # gather before the ring exists
        e.gather(np.zeros(8, np.int64))
    e.buffer_create(16)
    with pytest.raises(DsactError):       # empty ring (reference: np.random.randint(0, 0) raises too)
        e.gather(np.zeros(8, np.int64))
    e.buffer_add(np.zeros((4, 7), np.float32), np.zeros((4, 2), np.float32), np.zeros(4, np.float32),
                 np.zeros((4, 7), np.float32), np.zeros(4, np.float32))
    with pytest.raises(DsactError):       # index beyond `size`
        e.gather(np.full(8, 4, np.int64))
    with pytest.raises(DsactError):       # wrong batch
        e.gather(np.zeros(5, np.int64))
    with pytest.raises(DsactError):
        e.set_action_limits(np.zeros(2, np.float32), np.zeros(2, np.float32))
    with pytest.raises(DsactError):
        DsactEngine(7, 40, (32,), 8)      # act_dim > 32
    with pytest.raises(DsactError):
        e.graph_build(2)                   # no index table
    e.gather(np.array([0, 1, 2, 3, 3, 2, 1, 0], np.int64))
    e.step(0)
    assert all(np.isfinite(v) for v in e.read_stats().values())


def test_state_save_restore_continues_identically():
    """optimizer moments + Adam counters + mean_std EMA are not in the reference's checkpoints; the sidecar
    (dsact_get_state/set_state + the arenas) must resume the run bit-for-bit."""
    O, A, hid, B = 13, 3, (64, 64), 32
    a1, _ = make_pair(O, A, hid, B, seed=5)
    rng = np.random.default_rng(11)
    batches = [synth_batch(rng, B, O, A) for _ in range(6)]
    torch.manual_seed(21)
    noises = [draw_noise(B, A) for _ in range(6)]

    def run(alg, its):
        for it in its:
            d, n = batches[it], noises[it]
            alg.engine.load_batch(*(d[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            alg.engine.set_noise(n["eps_new"].numpy(), n["eps_2"].numpy(), n["z5"].numpy(), n["z6"].numpy())
            alg.engine.step(it)
        alg.engine.sync()

    run(a1, range(3))
    sd = {k: v.clone() for k, v in a1.networks.state_dict().items()}
    side = dict(a1.engine.get_state(), m=a1.engine.adam_m.clone(), v=a1.engine.adam_v.clone())
    run(a1, range(3, 6))
    a2, _ = make_pair(O, A, hid, B, seed=99)          # different init, then restored
    a2.networks.load_state_dict(sd)
    a2.engine.adam_m.copy_(side["m"]); a2.engine.adam_v.copy_(side["v"])
    torch.cuda.synchronize()
    a2.engine.set_state(side["adam_steps"], side["mean_std"])
    run(a2, range(3, 6))
    assert torch.equal(a1.engine.online, a2.engine.online)
    assert torch.equal(a1.engine.target, a2.engine.target)
    assert a1.engine.get_state() == a2.engine.get_state()


def test_optimizer_sidecar_roundtrip_resumes_bitwise(tmp_path):
    """f2 (optional in SURVEY 8f; absent in the reference, whose checkpoints hold the networks only): `optimizer_state_dict()`
    / `load_optimizer_state_dict()` through torch.save / torch.load + the reference-format network checkpoint resume a run bit
    for bit; a sidecar of another layout is refused; the trainer writes one next to every apprfunc file when asked to."""
    O, A, hid, B = 16, 4, (64, 64), 32
    a1, _ = make_pair(O, A, hid, B, seed=5)
    rng = np.random.default_rng(11)
    batches = [synth_batch(rng, B, O, A) for _ in range(6)]

    def run(alg, its):
        for it in its:
            torch.manual_seed(300 + it)
            alg.local_update(batches[it], it)
        alg.engine.sync()

    run(a1, range(3))
    torch.save(a1.networks.state_dict(), tmp_path / "apprfunc_3.pkl")
    torch.save(a1.optimizer_state_dict(), tmp_path / "apprfunc_3.optstate.pkl")
    run(a1, range(3, 6))
    a2, _ = make_pair(O, A, hid, B, seed=99)          # different init, then restored from the two files
    a2.networks.load_state_dict(torch.load(tmp_path / "apprfunc_3.pkl"))
    a2.load_optimizer_state_dict(torch.load(tmp_path / "apprfunc_3.optstate.pkl"))
    run(a2, range(3, 6))
    for n in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a1.engine, n), getattr(a2.engine, n)), n
    assert a1.engine.get_state() == a2.engine.get_state()
    a3, _ = make_pair(O, A, (64, 64, 64), B, seed=1)
    with pytest.raises(ValueError, match="another network layout"):
        a3.load_optimizer_state_dict(torch.load(tmp_path / "apprfunc_3.optstate.pkl"))
    from training.hip_trainer import HipOffSerialTrainer

    class Buf:
        size = 10 ** 6

        def sample_batch(self, n):
            return batches[0]

        def __get_RAM__(self):
            return 0.0

    tr = HipOffSerialTrainer(a2, None, Buf(), None, replay_batch_size=B, max_iteration=2, log_save_interval=10, apprfunc_save_interval=1,
                             eval_interval=10 ** 9, save_folder=str(tmp_path / "run"), buffer_warm_size=0, save_optimizer_state=True)
    tr.train()
    files = sorted(os.listdir(tmp_path / "run" / "apprfunc"))
    assert "apprfunc_1.pkl" in files and "apprfunc_1.optstate.pkl" in files and "apprfunc_2.optstate.pkl" in files
    side = torch.load(tmp_path / "run" / "apprfunc" / "apprfunc_2.optstate.pkl")
    assert side["iteration"] == 2 and side["format"] == a2.SIDECAR_FORMAT and side["adam_m"].numel() == a2.engine.adam_m.numel()


def test_full_size_replay_gather_and_determinism():
    """BASELINE.json configs[1] sizes: 1M-row ring in HBM, batch 256 -- gathered rows equal torch's
    index_select on the same device arrays (bit-exact), and two replays of the same 8 updates agree bit-for-bit."""
    from dsact.engine import DsactEngine

    O, A, B, N = 376, 17, 256, 1_000_000
    outs = []
    for rep in range(2):
        alg, _ = make_pair(O, A, (256, 256), B, seed=1)
        e = alg.engine
        e.set_device_rng(7)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(3)
        cols = {}
        for r0 in range(0, N, 250_000):
            n = 250_000
            obs = torch.randn(n, O, device="cuda", generator=g); obs2 = torch.randn(n, O, device="cuda", generator=g)
            act = torch.rand(n, A, device="cuda", generator=g) - .5; rew = torch.randn(n, device="cuda", generator=g)
            done = (torch.rand(n, device="cuda", generator=g) < .01).float()
            e.buffer_fill_device(r0, obs, act, rew, obs2, done)
            if r0 == 750_000:
                cols = dict(obs=obs, obs2=obs2, act=act, rew=rew, done=done)
        assert e.buffer_size == N and e.buffer_ptr == 0
        np.random.seed(5)
        idx = np.random.randint(750_000, N, size=B)
        e.gather(idx)
        got = e.read_batch(with_logp=False)
        ti = torch.as_tensor(idx - 750_000, device="cuda")
        for k in ("obs", "obs2", "act", "rew", "done"):
            assert np.array_equal(got[k], cols[k].index_select(0, ti).cpu().numpy()), k
        np.random.seed(6)
        e.upload_index_table(np.random.randint(0, N, size=(8, B)))
        e.graph_build(2)
        e.graph_run(0, 8)
        e.sync()
        outs.append((e.online.clone(), e.read_stats()))
        del alg, e, cols
        torch.cuda.empty_cache()
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def test_ten_million_row_ring_addressing():
    """BASELINE.json configs[4]'s single-GPU leg (training/replay_buffer.py:20-50 semantics at buffer_max_size = 10M): a
    10,000,000-row Humanoid ring is 30.9 GB, and row x obs_dim crosses 2^31 floats at row 5,711,393 -- every ring offset is
    size_t (csrc/dsact_kernels.h k_gather / k_ring_write), the indices travel as int32. Rows written at 0, around the 2^31
    crossing (device fill AND the ring-write kernel of dsact_buffer_add) and at 9,999,999 come back from the gather bit for bit;
    a graph of updates that samples them runs finite."""
    O, A, B, N = 376, 17, 1024, 10_000_000
    cross = (1 << 31) // O                      # 5,711,393: the first row whose LAST float sits past 2^31
    alg, _ = make_pair(O, A, (256, 256, 256), B, seed=1)
    e = alg.engine
    e.set_device_rng(7)
    e.buffer_create(N)
    g = torch.Generator(device="cuda").manual_seed(3)

    def rows(n):
        return dict(obs=torch.randn(n, O, device="cuda", generator=g), act=torch.rand(n, A, device="cuda", generator=g) - .5,
                    rew=torch.randn(n, device="cuda", generator=g), obs2=torch.randn(n, O, device="cuda", generator=g),
                    done=(torch.rand(n, device="cuda", generator=g) < .5).float())

    want = {}
    def put(r0, d):
        for i in range(d["rew"].shape[0]):
            want[r0 + i] = {k: v[i].cpu().numpy() for k, v in d.items()}

    # the last row first: the ring counts as full from here on (size = 10M, ptr = 0)
    d = rows(1); e.buffer_fill_device(N - 1, d["obs"], d["act"], d["rew"], d["obs2"], d["done"]); put(N - 1, d)
    assert e.buffer_size == N and e.buffer_ptr == 0
    d = rows(2); e.buffer_fill_device(0, d["obs"], d["act"], d["rew"], d["obs2"], d["done"]); put(0, d)
    d = rows(3); e.buffer_fill_device(cross - 2, d["obs"], d["act"], d["rew"], d["obs2"], d["done"]); put(cross - 2, d)
    assert e.buffer_ptr == cross + 1
    # the ring-write kernel (dsact_buffer_add -> k_ring_write) continues at ptr = cross + 1, on the far side of 2^31 floats
    d = {k: v.cpu() for k, v in rows(4).items()}
    e.buffer_add(d["obs"].numpy(), d["act"].numpy(), d["rew"].numpy(), d["obs2"].numpy(), d["done"].numpy())
    put(cross + 1, d)
    assert e.buffer_ptr == cross + 5 and e.buffer_size == N
    keys = sorted(want)
    idx = np.array([keys[i % len(keys)] for i in range(B)], dtype=np.int64)
    e.gather(idx)
    got = e.read_batch(with_logp=False)
    for b, r in enumerate(idx):
        for k in ("obs", "obs2", "act", "rew", "done"):
            assert np.array_equal(got[k][b], want[int(r)][k]), (int(r), k)
    # index rows past int32-sized offsets through the graph's device-side gather too
    tab = np.stack([np.roll(idx, s) for s in range(4)])
    e.upload_index_table(tab)
    e.graph_build(2)
    e.graph_run(0, 4)
    e.sync()
    assert torch.isfinite(e.online).all() and all(np.isfinite(v) for v in e.read_stats().values())
    got = e.read_batch(with_logp=False)          # the staged minibatch = the last table row
    for b, r in enumerate(tab[3]):
        assert np.array_equal(got["obs2"][b], want[int(r)]["obs2"]), int(r)
    with pytest.raises(Exception):
        e.gather(np.full(B, N, dtype=np.int64))   # one past the end is refused, not wrapped
    del alg, e
    torch.cuda.empty_cache()


def test_local_update_takes_cuda_tensors_like_the_reference_trainer():
    """The reference trainer hands `local_update` per-key `.cuda()` tensors (training/trainer.py:72-74).
    dsact_load_batch takes them by device address: the staged rows are bit-identical to staging the CPU batch,
    and so is the update that follows (mixed CPU/CUDA dicts included)."""
    O, A, hid, B = 23, 5, (64, 96, 64), 64
    a_cpu, _ = make_pair(O, A, hid, B, seed=4)
    a_gpu, _ = make_pair(O, A, hid, B, seed=4)
    a_mix, _ = make_pair(O, A, hid, B, seed=4)
    rng = np.random.default_rng(8)
    for it in range(3):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        on_gpu = {k: v.cuda() for k, v in data.items()}
        mixed = {k: (v.cuda() if k in ("obs", "rew") else v) for k, v in data.items()}
        for alg, d in ((a_cpu, data), (a_gpu, on_gpu), (a_mix, mixed)):
            torch.manual_seed(100 + it)     # strict_rng: the same 8 host draws for each
            alg.local_update(d, it)
        want = a_cpu.engine.read_batch(with_logp=False)
        for alg in (a_gpu, a_mix):
            got = alg.engine.read_batch(with_logp=False)
            for k in ("obs", "act", "rew", "obs2", "done"):
                assert np.array_equal(got[k], want[k]), k
    for alg in (a_cpu, a_gpu, a_mix):
        alg.engine.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a_cpu.engine, name), getattr(a_gpu.engine, name)), name
        assert torch.equal(getattr(a_cpu.engine, name), getattr(a_mix.engine, name)), name


class _ModuleOnDevice:
    """what the reference's trainer / evaluator wrap every sampling call in (utils/common_utils.py:164-177:
    remember the device of the first parameter, `.to(new)` on enter, `.to(previous)` on exit), restated for the test"""

    def __init__(self, module, device):
        self.module, self.new = module, device
        self.prev = next(module.parameters()).device.type
        self.moved = self.prev != device

    def __enter__(self):
        if self.moved:
            self.module.to(self.new)

    def __exit__(self, *exc):
        if self.moved:
            self.module.to(self.prev)


def test_attached_container_survives_the_reference_device_ping_pong():
    """the reference trainer moves `networks` to the CPU around every sampler / evaluator
    call and back (training/trainer.py:63-66, ModuleOnDevice). With the container attached to the engine the
    parameters ARE the HIP arenas: `.to("cpu")`, `.cpu()`, `.cuda()`, the context manager must neither re-home them
    nor break acting with CPU observations, and training must continue on the same storage."""
    O, A, B = 11, 3, 64
    alg, orc = make_pair(O, A, (64, 64), B, seed=8)
    nets, e = alg.networks, alg.engine
    ptr0 = [p.data_ptr() for p in nets.parameters()]
    lo, hi = e.online.data_ptr(), e.online.data_ptr() + 4 * e.online.numel()
    assert all(p.is_cuda for p in nets.parameters())
    rng = np.random.default_rng(4)
    obs1 = torch.as_tensor(rng.standard_normal((1, O), dtype=np.float32))

    def act_logits():
        with torch.no_grad():
            lg = nets.policy(obs1)                 # CPU observation in, CPU logits out (off_sampler.py:44-51)
        assert lg.device.type == "cpu" and lg.shape == (1, 2 * A)
        dist = nets.create_action_distributions(lg)
        a, lp = dist.sample()
        assert a.shape == (1, A) and lp.shape == (1,)
        return lg

    before = act_logits()
    for move in (lambda: nets.to("cpu"), lambda: nets.cpu(), lambda: nets.cuda(), lambda: nets.to("cuda:0"),
                 lambda: nets.to(torch.device("cpu"))):
        out = move()
        assert out is nets
        assert [p.data_ptr() for p in nets.parameters()] == ptr0
        assert all(p.is_cuda and lo <= p.data_ptr() < hi for n, p in nets.named_parameters() if "target" not in n)
    with _ModuleOnDevice(nets, "cpu") as _:
        inside = act_logits()                       # the sampler's view of the module
        assert [p.data_ptr() for p in nets.parameters()] == ptr0
    assert torch.equal(before, inside)
    # the logits are the live learner weights: identical to the oracle's torch forward of the same parameters
    with torch.no_grad():
        want = orc._pi(obs1, orc.p["policy"])
    assert torch.allclose(before, want, atol=2e-5, rtol=1e-5)
    # training continues on the same storage, and the next acting call sees the UPDATED weights
    for it in range(3):
        data = synth_batch(rng, B, O, A)
        torch.manual_seed(60 + it)
        noise = draw_noise(B, A)
        torch.manual_seed(60 + it)
        with _ModuleOnDevice(nets, "cpu"):
            pass                                    # trainer.py:63-66 around sampler.sample()
        tb = alg.local_update({k: v.cuda() for k, v in data.items()}, it)   # trainer.py:72-74
        ref = orc.local_update(data, noise, it)
        assert abs(float(tb["Loss/Critic loss-RL iter"]) - float(ref["Loss/Critic loss-RL iter"])) <= 1e-5 * max(1.0, abs(float(ref["Loss/Critic loss-RL iter"])))
    after = act_logits()
    assert not torch.equal(after, before)
    with torch.no_grad():
        want = orc._pi(obs1, orc.p["policy"])
    assert torch.allclose(after, want, atol=2e-5, rtol=1e-5)
    assert [p.data_ptr() for p in nets.parameters()] == ptr0
    # state_dict round trip through the CPU (the reference's save / sampler.load_state_dict path)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in nets.state_dict().items()}
    nets.load_state_dict(sd_cpu)
    assert [p.data_ptr() for p in nets.parameters()] == ptr0
    assert torch.equal(act_logits(), after)


@pytest.mark.parametrize("O,A,hid,B", [(376, 17, (256, 256, 256), 256), (24, 6, (128, 128), 64), (12, 3, (64, 64), 16)])
def test_merged_forward_launch_equals_two_launches(O, A, hid, B, monkeypatch):
    """k_chain_fwd2 (forward groups A and B in one launch, group B waiting on per-slice ready flags) == the two
    launches it replaces (DSACT_NO_FWD_MERGE=1), bit for bit over 6 updates -- eager steps and a graph replay; and the
    spin-timeout word stays clear (statistics finite)."""
    algs = []
    for merged in (True, False):
        if merged:
            monkeypatch.delenv("DSACT_NO_FWD_MERGE", raising=False)
        else:
            monkeypatch.setenv("DSACT_NO_FWD_MERGE", "1")
        alg, _ = make_pair(O, A, hid, B, seed=21)
        assert alg.engine.chain_active
        algs.append(alg)
    monkeypatch.delenv("DSACT_NO_FWD_MERGE", raising=False)
    rng = np.random.default_rng(12)
    for it in range(3):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(300 + it)
        noise = draw_noise(B, A)
        for a in algs:
            a.engine.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            a.engine.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
            a.engine.step(it)
    N = 2048
    for a in algs:
        e = a.engine
        e.set_device_rng(99)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(4)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(2)
        e.upload_index_table(np.random.randint(0, N, size=(4, B)))
        e.graph_build(3)
        e.graph_run(3, 3)
        e.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(algs[0].engine, name), getattr(algs[1].engine, name)), name
    st0, st1 = algs[0].engine.read_stats(), algs[1].engine.read_stats()
    st0.pop("_device_ms"); st1.pop("_device_ms")   # a timing, not a statistic
    # policy_mean / policy_std are sums of per-slice partials: the merged launch runs the policy in 4-row slices, the two
    # launches in 8-row ones -- same values, another summation order
    for k in ("DSAC2/policy_mean-RL iter", "DSAC2/policy_std-RL iter"):
        a0, a1 = st0.pop(k), st1.pop(k)
        assert abs(a0 - a1) <= 1e-6 * max(abs(a1), 1e-3), (k, a0, a1)
    assert st0 == st1 and all(np.isfinite(v) for v in st0.values())
    names = [[k for k, _, _ in a.engine.profile_step(6)] for a in algs]
    assert "chain_fwd" in names[0] and "chain_fwd_a" not in names[0]
    assert "chain_fwd_a" in names[1] and "chain_fwd_b" in names[1] and "chain_fwd" not in names[1]


@pytest.mark.parametrize("O,A,hid,B", [(376, 17, (256, 256, 256), 256), (24, 6, (128, 128), 64), (12, 3, (64, 64), 16)])
def test_merged_policy_backward_equals_two_launches(O, A, hid, B, monkeypatch):
    """k_chain_bwd_pi with merge_dw (the policy's weight-gradient / Adam tiles and the closing block in the policy-backward
    launch, waiting for the chain's arrival counter) == policy backward + k_dw2 as two launches (DSACT_NO_PI_MERGE=1), bit
    for bit: gradients of the unfused entry point, then 4 fused eager updates (both delayed-update parities) and a graph
    replay; every statistic equal; the hand-off word stays clear."""
    algs = []
    for merged in (True, False):
        if merged:
            monkeypatch.delenv("DSACT_NO_PI_MERGE", raising=False)
        else:
            monkeypatch.setenv("DSACT_NO_PI_MERGE", "1")
        alg, _ = make_pair(O, A, hid, B, seed=23)
        assert alg.engine.chain_active and alg.engine.debug_get("pi_merge") == (1.0 if merged else 0.0)
        algs.append(alg)
    monkeypatch.delenv("DSACT_NO_PI_MERGE", raising=False)
    rng = np.random.default_rng(13)
    for it in range(5):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(400 + it)
        noise = draw_noise(B, A)
        for a in algs:
            e = a.engine
            e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
            if it == 0:
                e.compute_grads(it)
                e.sync()
                e.apply_update(it)
            else:
                e.step(it)
        if it == 0:
            assert torch.equal(algs[0].engine.grads, algs[1].engine.grads)
            assert bool(torch.isfinite(algs[0].engine.grads).all())
    N = 2048
    for a in algs:
        e = a.engine
        _fill_ring(e, N, O, A, 4)
        e.set_device_rng(98)
        np.random.seed(3)
        e.upload_index_table(np.random.randint(0, N, size=(4, B)))
        e.graph_build(4)
        e.graph_run(5, 4)
        e.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(algs[0].engine, name), getattr(algs[1].engine, name)), name
    st0, st1 = algs[0].engine.read_stats(), algs[1].engine.read_stats()
    st0.pop("_device_ms"); st1.pop("_device_ms")
    assert st0 == st1 and all(np.isfinite(v) for v in st0.values())
    assert algs[0].engine.get_state() == algs[1].engine.get_state()
    names = [[k for k, _, _ in a.engine.profile_step(9)] for a in algs]
    assert "dW" not in names[0] and "dW" in names[1] and "chain_bwd_pi" in names[0]
    assert algs[0].engine.debug_get("handoff_failures") == 0.0


def _fill_ring(e, N, O, A, seed):
    e.buffer_create(N)
    g = torch.Generator(device="cuda").manual_seed(seed)
    e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                         torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                         (torch.rand(N, device="cuda", generator=g) < .05).float())


def test_handover_buffers_poisoned_between_updates():
    """ADVICE r2 (ordering of the in-launch hand-over): everything the merged launches hand from producer to consumer
    workgroups (forward: saved observation parts `zobs`, the sampled-action columns, dL/da; policy backward: the policy's
    dZ packs its weight-gradient tiles wait for) is filled with NaN before EVERY update. A consumer that passed its flag before the producer's stores had landed -- or that read a stale cached
    copy -- would compute on NaN. 40 updates at the BASELINE shape: merged == two launches bit for bit, all finite."""
    O, A, hid, B = 376, 17, (256, 256, 256), 256
    algs = [make_pair(O, A, hid, B, seed=31)[0] for _ in range(2)]
    algs[1].engine.debug_set("fwd_merge", 0)
    algs[1].engine.debug_set("pi_merge", 0)
    assert algs[0].engine.debug_get("fwd_merge") == 1.0 and algs[1].engine.debug_get("fwd_merge") == 0.0
    assert algs[0].engine.debug_get("pi_merge") == 1.0 and algs[1].engine.debug_get("pi_merge") == 0.0
    rng = np.random.default_rng(14)
    for it in range(40):
        data = synth_batch(rng, B, O, A, p_done=0.1)
        torch.manual_seed(500 + it)
        noise = draw_noise(B, A)
        for a in algs:
            e = a.engine
            e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
            e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
            e.debug_set("poison_handover", float("nan"))
            e.step(it)
    for name in ("online", "target", "adam_m", "adam_v"):
        t0, t1 = getattr(algs[0].engine, name), getattr(algs[1].engine, name)
        assert bool(torch.isfinite(t0).all()), name
        assert torch.equal(t0, t1), name
    st = algs[0].engine.read_stats()
    assert all(np.isfinite(v) for k, v in st.items())
    assert algs[0].engine.debug_get("handoff_failures") == 0.0


@pytest.mark.parametrize("which", [1, 2, 3])
def test_forced_handover_timeout_fails_the_call_and_falls_back(which):
    """a consumer that gives up waiting must FAIL the call, not poison the statistics. One producer
    withholds its ready flag ("withhold_flag" 1: a forward unit's; 2: a policy-backward slice never arrives; 3: a slice of
    q1's backward chain never arrives in the merged critic-backward launch k_chain_bwd_qt, round 5): the next
    synchronising entry point returns DSACT_E_HIP, the handle drops to the unmerged launches (and re-captures its graph
    without the merged ones), and from restored state it trains on, bit-identical to an engine that never merged."""
    from dsact._ffi import DsactError

    O, A, hid, B, N = 24, 6, (128, 128), 64, 1024
    alg, _ = make_pair(O, A, hid, B, seed=41)
    ref, _ = make_pair(O, A, hid, B, seed=41)
    ref.engine.debug_set("fwd_merge", 0)
    ref.engine.debug_set("pi_merge", 0)
    e, r = alg.engine, ref.engine
    assert e.debug_get("fwd_merge") == 1.0 and e.debug_get("pi_merge") == 1.0
    for x in (e, r):
        _fill_ring(x, N, O, A, 8)
        x.set_device_rng(5)
        np.random.seed(4)
        x.upload_index_table(np.random.randint(0, N, size=(8, B)))
    snap = {k: v.clone() for k, v in alg.networks.state_dict().items()}
    arenas = {n: getattr(e, n).clone() for n in ("adam_m", "adam_v")}
    state = e.get_state()
    # --- a graph replay with a withheld flag: the launch succeeds, the first synchronising call fails
    first = 1 if which == 3 else 0    # (the merged critic backward belongs to an update that leaves the policy alone and has a successor)
    e.debug_set("withhold_flag", which)
    e.graph_build(2)
    e.graph_run(first, 2)             # asynchronous: returns before the consumers give up
    with pytest.raises(DsactError, match="hand-over timed out"):
        e.sync()
    assert e.debug_get("handoff_failures") == 1.0 and e.debug_get("fwd_merge") == 0.0 and e.debug_get("pi_merge") == 0.0
    assert e.debug_get("graph_steps") == 2.0     # captured again, without the merged launches
    e.sync()                                      # the word was consumed: no second error
    # the fused Adam / Polyak epilogues ran on whatever the consumers found -- the handle refuses to train on
    # until the caller has restored the state and acknowledged
    assert e.debug_get("state_invalid") == 1.0
    for call in (lambda: e.step(0), lambda: e.graph_run(first, 2), lambda: e.profile_step(0)):
        with pytest.raises(DsactError, match="invalid after a hand-over timeout"):
            call()
    # --- restore the state the failed call invalidated, then both engines run the same updates (the reference engine
    #     replays the same updates first so that both index-table cursors agree, and is restored the same way)
    r.graph_build(2)
    r.graph_run(first, 2)
    r.sync()
    for a_, x in ((alg, e), (ref, r)):
        a_.networks.load_state_dict(snap)
        for n, t in arenas.items():
            getattr(x, n).copy_(t)
        torch.cuda.synchronize()
        x.set_state(adam_steps=state["adam_steps"], mean_std=state["mean_std"])
    assert e.debug_get("state_invalid") == 0.0
    names = [k for k, _, _ in e.profile_step(0)]
    assert "chain_fwd_a" in names and "dW" in names
    r.profile_step(0)
    for x in (e, r):
        x.graph_run(0, 4)
        x.sync()
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(e, name), getattr(r, name)), name
    st = e.read_stats()
    assert all(np.isfinite(v) for v in st.values())
    # eager path after the fallback too
    with pytest.raises(DsactError):
        e.debug_set("no_such_switch", 1)


def test_repeated_forward_half_clears_its_ready_flags():
    """dsact_dp_enqueue_forward may be called twice (or never be followed by its backward); the second merged
    forward must not find the first one's ready flags raised. Two forwards + backward == one forward + backward."""
    O, A, hid, B, N = 24, 6, (128, 128), 64, 512
    outs = []
    for reps in (1, 2):
        alg, _ = make_pair(O, A, hid, B, seed=43)
        e = alg.engine
        assert e.debug_get("fwd_merge") == 1.0
        _fill_ring(e, N, O, A, 9)
        np.random.seed(6)
        e.upload_index_table(np.random.randint(0, N, size=(2, B)))
        torch.manual_seed(8)
        noise = draw_noise(B, A)
        e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
        e.dp_set_strict(True)
        e.dp_begin(0)
        for k in range(reps):
            e.dp_forward()
            if k + 1 < reps:
                e.debug_set("poison_handover", float("nan"))
        e.sync()
        e.dp_backward()
        e.sync()
        outs.append(e.grads.clone())
        assert e.debug_get("handoff_failures") == 0.0
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
