"""The scalar closed forms every HIP kernel uses (dsac-v2_amd/csrc/dsact_math.h), compiled for the
host, against torch ops / autograd -- the same ops the reference calls. CPU only."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def hm():
    import __graft_entry__ as g
    g.build()
    return C.CDLL(os.path.join(ROOT, "oracle", "_build", "libdsact_hostmath.so"))


def p(a):
    return a.ctypes.data_as(FP)


def f32(t):
    return np.ascontiguousarray(t.detach().numpy().astype(np.float32))


def test_gelu_and_softplus(hm):
    torch.manual_seed(0)   # the draws decide whether the 1-2 ulp comparisons below meet a large |z|: keep them fixed
    z = torch.cat([torch.randn(5000) * 3, torch.tensor([0.0, -10.0, 10.0, 25.0, -25.0])]).requires_grad_(True)
    h = F.gelu(z)
    h.sum().backward()
    zn = f32(z)
    ho, go = np.empty_like(zn), np.empty_like(zn)
    hm.hm_gelu(p(zn), zn.size, p(ho), p(go))
    np.testing.assert_allclose(ho, f32(h), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(go, f32(z.grad), atol=2e-6, rtol=1e-6)
    x = torch.cat([torch.randn(3000) * 8, torch.tensor([20.0, 20.0001, 19.9999, -30.0])]).requires_grad_(True)
    y = F.softplus(x)
    y.sum().backward()
    xn = f32(x)
    yo, dyo = np.empty_like(xn), np.empty_like(xn)
    hm.hm_softplus(p(xn), xn.size, p(yo), p(dyo))
    np.testing.assert_allclose(yo, f32(y), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(dyo, f32(x.grad), atol=1e-6, rtol=1e-6)


def test_tanh_gauss_forward_backward(hm):
    from oracle.dsact_oracle import tanh_gauss_rsample
    n = 4000
    torch.manual_seed(0)
    mu = (torch.randn(n) * 1.5).requires_grad_(True)
    raw = torch.cat([torch.randn(n - 4) * 2 - 1, torch.tensor([0.5, 0.6, -20.0, -21.0])]).requires_grad_(True)
    eps = torch.randn(n)
    std = torch.clamp(raw, -20.0, 0.5).exp()
    hi, lo = torch.tensor([0.4]), torch.tensor([-0.4])
    # one action dimension per row: Independent(...).sum(-1) over a single column
    act, lp = tanh_gauss_rsample(torch.stack([mu, std], -1).unsqueeze(1).reshape(n, 2), eps.unsqueeze(-1), hi, lo)
    gA = torch.randn(n)
    gLp = 0.0123
    ((act.squeeze(-1) * gA).sum() + gLp * lp.sum()).backward()
    a_o, lp_o = np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_tanh_gauss_fwd.argtypes = [FP, FP, FP, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, FP, FP]
    hm.hm_tanh_gauss_fwd(p(f32(mu)), p(f32(raw)), p(f32(eps)), n, 0.4, 0.0, -20.0, 0.5, p(a_o), p(lp_o))
    np.testing.assert_allclose(a_o, f32(act.squeeze(-1)), atol=1e-6)
    np.testing.assert_allclose(lp_o, f32(lp), atol=3e-5, rtol=1e-5)
    dmu, draw = np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_tanh_gauss_bwd.argtypes = [FP, FP, FP, FP, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, FP, FP]
    hm.hm_tanh_gauss_bwd(p(f32(mu)), p(f32(raw)), p(f32(eps)), p(f32(gA)), n, 0.4, -20.0, 0.5, gLp, p(dmu), p(draw))
    np.testing.assert_allclose(dmu, f32(mu.grad), atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(draw, f32(raw.grad), atol=2e-6, rtol=2e-4)


def test_critic_term_matches_reference_loss_autograd(hm):
    n = 5000
    torch.manual_seed(1)
    q = (torch.randn(n) * 40).requires_grad_(True)      # many |q - tq| > 50 (Huber saturated)
    std = (torch.rand(n) * 3 + 1e-3).requires_grad_(True)
    tq, tqs = torch.randn(n) * 40, torch.randn(n) * 60
    ms = torch.tensor(0.8)
    sd = torch.clamp(std, min=0.0).detach()
    ratio = (ms.pow(2) / (sd.pow(2) + 0.1)).clamp(min=0.1, max=10)
    tqb = (q.detach() + torch.clamp(tqs - q.detach(), -3 * ms, 3 * ms))
    hub = lambda a, b: F.huber_loss(a, b, delta=50, reduction="none")
    per = ratio * (hub(q, tq) + std * (sd.pow(2) - hub(q.detach(), tqb)) / (sd + 0.1))
    per.sum().backward()
    lo, dq, ds = (np.empty(n, np.float32) for _ in range(3))
    hm.hm_critic.argtypes = [FP, FP, FP, FP, C.c_int, C.c_float, FP, FP, FP]
    hm.hm_critic(p(f32(q)), p(f32(std)), p(f32(tq)), p(f32(tqs)), n, 0.8, p(lo), p(dq), p(ds))
    np.testing.assert_allclose(lo, f32(per), rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(dq, f32(q.grad), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(ds, f32(std.grad), rtol=2e-6, atol=1e-5)


def test_adam_and_polyak_track_torch(hm):
    n = 4096
    torch.manual_seed(2)
    prm = torch.randn(n).requires_grad_(True)
    opt = torch.optim.Adam([prm], lr=1e-4)
    pn, m, v = f32(prm).copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    hm.hm_adam.argtypes = [FP, FP, FP, FP, C.c_int] + [C.c_float] * 6
    tgt = torch.randn(n)
    tn = f32(tgt).copy()
    hm.hm_polyak.argtypes = [FP, FP, C.c_int, C.c_float, C.c_float]
    for t in range(1, 8):
        g = torch.randn(n) * (10.0 ** float(np.random.default_rng(t).integers(-6, 2)))
        prm.grad = g.clone()
        opt.step()
        ss = np.float32(1e-4 / (1 - 0.9 ** t))
        bc2 = np.float32((1 - 0.999 ** t) ** 0.5)
        hm.hm_adam(p(pn), p(m), p(v), p(f32(g)), n, np.float32(1 - 0.9), np.float32(0.999), np.float32(1 - 0.999), ss, bc2, np.float32(1e-8))
        np.testing.assert_allclose(pn, f32(prm), atol=2e-9, rtol=3e-7)
        polyak = 1 - 0.005
        tgt.mul_(polyak)
        tgt.add_((1 - polyak) * prm.data)
        hm.hm_polyak(p(tn), p(pn), n, np.float32(polyak), np.float32(1 - polyak))
        np.testing.assert_allclose(tn, f32(tgt), atol=1e-9, rtol=3e-7)
    st = opt.state[prm]
    np.testing.assert_allclose(m, f32(st["exp_avg"]), rtol=1e-6, atol=1e-7 * float(np.abs(m).max()))
    np.testing.assert_allclose(v, f32(st["exp_avg_sq"]), rtol=1e-6, atol=1e-7 * float(np.abs(v).max()))


def test_gauss_distribution_branch(hm):
    """s == 0 selects the reference's plain GaussDistribution (utils/act_distribution_cls.py:82-115) in the closed forms:
    action = mu + sigma * eps, log-prob of the diagonal Gaussian, and its gradients against torch autograd."""
    n = 3000
    torch.manual_seed(1)
    mu = (torch.randn(n) * 1.5).requires_grad_(True)
    raw = torch.cat([torch.randn(n - 4) * 2 - 1, torch.tensor([0.5, 0.6, -20.0, -21.0])]).requires_grad_(True)
    eps = torch.randn(n)
    std = torch.clamp(raw, -20.0, 0.5).exp()
    base = torch.distributions.Normal(mu, std)
    act = mu + eps * std                      # Normal.rsample
    lp = base.log_prob(act)
    gA = torch.randn(n)
    gLp = 0.37
    (act * gA).sum().backward(retain_graph=True)
    (lp * gLp).sum().backward()
    mun, rawn, epsn, gAn = f32(mu), f32(raw), f32(eps), f32(gA)
    a_o, lp_o = np.empty_like(mun), np.empty_like(mun)
    hm.hm_tanh_gauss_fwd.argtypes = [FP, FP, FP, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, FP, FP]
    hm.hm_tanh_gauss_fwd(p(mun), p(rawn), p(epsn), n, 0.0, 0.0, -20.0, 0.5, p(a_o), p(lp_o))
    np.testing.assert_allclose(a_o, f32(act), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(lp_o, f32(lp), atol=2e-5, rtol=1e-5)
    dmu, draw = np.empty_like(mun), np.empty_like(mun)
    hm.hm_tanh_gauss_bwd.argtypes = [FP, FP, FP, FP, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, FP, FP]
    hm.hm_tanh_gauss_bwd(p(mun), p(rawn), p(epsn), p(gAn), n, 0.0, -20.0, 0.5, gLp, p(dmu), p(draw))
    # (autograd's d logp / d mu is the difference of two equal terms: zero up to their rounding)
    np.testing.assert_allclose(dmu, f32(mu.grad), atol=2e-5 * float(np.abs(f32(mu.grad)).max()) + 1e-5, rtol=0)
    np.testing.assert_allclose(draw, f32(raw.grad), atol=2e-5 * float(np.abs(f32(raw.grad)).max()) + 1e-5, rtol=1e-5)



@pytest.mark.parametrize("O,A,hid,act", [(376, 17, (256, 256, 256), "gelu"), (3, 1, (64, 64), "gelu"), (23, 5, (96, 40), "gelu"),
                                         (11, 3, (33,), "tanh"), (16, 4, (64, 64, 64, 64), "relu")])
def test_host_acting_forward_and_sampling_step(hm, O, A, hid, act):
    """csrc/dsact_host_act.h -- the sampler's batch-1 policy forward + TanhGaussDistribution.sample() on the host (SURVEY 8 f1;
    reference training/off_sampler.py:46-56, networks/mlp.py:79-100, utils/act_distribution_cls.py:32-42) -- against the torch
    CPU module the reference acts with: logits within fp32 summation-order noise, the action and log-prob of the same N(0,1)
    draw at the gates of the GPU acting forward's test. Every instantiation the CPU offers (512-bit, 256-bit FMA, baseline x86-64) and
    the fork-join pool (3 / 4 threads: bitwise the single-thread result)."""
    import torch.nn as nn

    acts = {"gelu": (0, nn.GELU), "relu": (1, nn.ReLU), "tanh": (5, nn.Tanh)}
    torch.manual_seed(5)
    sizes = [O] + list(hid) + [2 * A]
    layers = []
    for j in range(len(sizes) - 1):
        layers += [nn.Linear(sizes[j], sizes[j + 1]), acts[act][1]() if j < len(sizes) - 2 else nn.Identity()]
    net = nn.Sequential(*layers)
    flat, w_off, b_off, k_in, n_out = [], [], [], [], []
    off = 0
    for m in net:
        if isinstance(m, nn.Linear):
            w_off.append(off); flat.append(f32(m.weight).reshape(-1)); off += m.weight.numel()
            b_off.append(off); flat.append(f32(m.bias)); off += m.bias.numel()
            k_in.append(m.in_features); n_out.append(m.out_features)
    params = np.concatenate(flat)
    L = len(k_in)
    lim, lo_ls, hi_ls = 0.4, -20.0, 0.5
    scale, center = np.full(A, lim, np.float32), np.zeros(A, np.float32)
    rng = np.random.default_rng(1)
    IA, LA = (C.c_int * L), (C.c_longlong * L)
    for i in range(12):
        obs = (2.0 * rng.standard_normal(O)).astype(np.float32)
        with torch.no_grad():
            raw = net(torch.from_numpy(obs)[None].double().float())
            raw64 = net.double()(torch.from_numpy(obs)[None].double()); net.float()
        mean, std = raw[0, :A], torch.clamp(raw[0, A:], lo_ls, hi_ls).exp()
        torch.manual_seed(100 + i)
        eps = torch.randn(1, A)
        hm.hm_policy_act.argtypes = [FP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                     C.c_int, FP, C.c_int, C.c_float, C.c_float, FP, FP, FP, FP, FP, C.c_int, C.c_int, C.POINTER(C.c_int)]
        per_isa = {}
        for isa, threads in ((0, 1), (1, 1), (2, 1), (-1, 3), (0, 4)):
            logits = np.empty(2 * A, np.float32)
            rc = hm.hm_policy_act(p(params), L, IA(*k_in), IA(*n_out), LA(*w_off), LA(*b_off), acts[act][0], p(obs), A, lo_ls, hi_ls,
                                  None, p(scale), p(center), p(logits), None, isa, threads, None)
            if rc == -2:
                continue                     # this CPU lacks the instruction set
            assert rc >= 0
            # against the fp64 forward: both fp32 evaluations (torch's and this one) sit within summation-order noise of it
            scale64 = float(raw64.abs().max()) + 1.0
            assert np.abs(logits[:A] - raw64[0, :A].numpy()).max() <= 2e-6 * scale64
            np.testing.assert_allclose(logits[:A], f32(mean), atol=4e-6 * scale64, rtol=0)
            np.testing.assert_allclose(logits[A:], f32(std), rtol=2e-5, atol=1e-7)
            # the thread count never changes a bit (an output row is one thread's, in one order); the vector width may
            key = rc if isa < 0 else isa
            if key in per_isa:
                assert np.array_equal(per_isa[key], logits), (isa, threads)
            per_isa[key] = logits.copy()
            action, logp = np.empty(A, np.float32), np.empty(1, np.float32)
            hm.hm_policy_act(p(params), L, IA(*k_in), IA(*n_out), LA(*w_off), LA(*b_off), acts[act][0], p(obs), A, lo_ls, hi_ls,
                             p(f32(eps[0])), p(scale), p(center), p(action), p(logp), isa, threads, None)
            # the reference's sampling step on ITS logits and the same draw (act_distribution_cls.py:32-42)
            x = mean + std * eps[0]
            a_ref = lim * torch.tanh(x)
            lp_ref = (torch.distributions.Normal(mean, std).log_prob(x) - torch.log(1 + 1e-6 - torch.tanh(x) ** 2) - np.log(lim)).sum()
            np.testing.assert_allclose(action, f32(a_ref), atol=4e-6 * lim * scale64, rtol=0)
            t2 = (np.asarray(a_ref, np.float64) / lim) ** 2
            tol = 5e-4 + float((2.4e-7 / (1.0 + 1e-6 - np.minimum(t2, 1.0))).sum())
            assert abs(float(logp[0]) - float(lp_ref)) <= tol, (i, float(logp[0]), float(lp_ref), tol)
        assert len(per_isa) >= 2


def test_host_acting_forward_of_a_twin_trunk_policy(hm):
    """csrc/dsact_host_act.h on the arena's twin-trunk layout (a hidden layer = two row ranges with their own input halves) against
    the two torch MLPs, every instruction set the CPU offers and the fork-join pool"""
    import torch.nn as nn
    from dsact.layout import ArenaLayout

    O, A, hid = 23, 5, (96, 40, 64)
    torch.manual_seed(7)

    def mlp():
        sizes = [O] + list(hid) + [A]
        layers = []
        for j in range(len(sizes) - 1):
            layers += [nn.Linear(sizes[j], sizes[j + 1]), nn.GELU() if j < len(sizes) - 2 else nn.Identity()]
        return nn.Sequential(*layers)

    mean, lstd = mlp(), mlp()
    lay = ArenaLayout(O, A, list(hid), policy_std_type="mlp_separated")
    base = lay.net_offset["policy"][1]
    arena = torch.zeros(lay.n_online)
    named = {("mean.%s" % n): t for n, t in mean.named_parameters()}
    named.update({("log_std.%s" % n): t for n, t in lstd.named_parameters()})
    for name, _, off, shape, strides in lay.param_views("policy"):
        torch.as_strided(arena, shape, strides, off).copy_(named[name].detach())
    params = arena[base:base + lay.n_pi].numpy().copy()
    L = len(hid) + 1
    widths = [O] + [2 * h for h in hid]
    k_in, n_out, half, w_off, b_off = [], [], [], [], []
    off = 0
    for l in range(L):
        blk = 0 < l < L - 1
        n = 2 * hid[l] if l < L - 1 else 2 * A
        k = widths[l] // 2 if blk else widths[l]
        k_in.append(k); n_out.append(n); half.append(hid[l] if blk else 0)
        w_off.append(off); off += n * k if (blk or l == 0) else n * widths[l]
        b_off.append(off); off += n
    assert off == lay.n_pi
    lo_ls, hi_ls = -20.0, 0.5
    scale, center = np.full(A, 0.4, np.float32), np.zeros(A, np.float32)
    IA, LA = (C.c_int * L), (C.c_longlong * L)
    hm.hm_policy_act.argtypes = [FP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.c_int, FP, C.c_int, C.c_float, C.c_float, FP, FP, FP, FP, FP, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(2)
    for _ in range(6):
        obs = (2.0 * rng.standard_normal(O)).astype(np.float32)
        with torch.no_grad():
            m = mean(torch.from_numpy(obs)[None])[0]
            s = torch.clamp(lstd(torch.from_numpy(obs)[None])[0], lo_ls, hi_ls).exp()
        seen = {}
        for isa, threads in ((0, 1), (1, 1), (2, 1), (-1, 3), (0, 4)):
            logits = np.empty(2 * A, np.float32)
            rc = hm.hm_policy_act(p(params), L, IA(*k_in), IA(*n_out), LA(*w_off), LA(*b_off), 0, p(obs), A, lo_ls, hi_ls,
                                  None, p(scale), p(center), p(logits), None, isa, threads, IA(*half))
            if rc == -2:
                continue
            assert rc >= 0
            sc = float(m.abs().max()) + 1.0
            np.testing.assert_allclose(logits[:A], f32(m), atol=4e-6 * sc, rtol=0)
            np.testing.assert_allclose(logits[A:], f32(s), rtol=2e-5, atol=1e-7)
            key = rc if isa < 0 else isa
            if key in seen:
                assert np.array_equal(seen[key], logits)
            seen[key] = logits.copy()
        assert len(seen) >= 2
