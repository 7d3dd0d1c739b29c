"""DSAC_V1_HIP (reference dsac_v1.py on the shared kernels) against the V1 oracle
(oracle/dsac_v1_oracle.py, pinned bit-exact to the live reference) and the reference golden vectors."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, hip_kwargs, synth_batch
from oracle.dsact_oracle import default_config
from oracle.dsac_v1_oracle import V1_TB_KEYS, DsacV1Oracle, draw_noise_v1
from test_hip_parity import AdamNoise, Report

pytestmark = pytest.mark.gpu


def make_pair(O, A, hid, B, act_limit=0.4, seed=0, init=None, td_bound=10.0, **over):
    from dsac_v1_hip import DSAC_V1_HIP

    torch.manual_seed(seed)
    alg = DSAC_V1_HIP(**hip_kwargs(O, A, hid, B, act_limit=act_limit, strict_rng=True, algorithm="DSAC_V1_HIP",
                                   TD_bound=td_bound, **over))
    if init is not None:
        alg.networks.load_state_dict(init)
    cfg = default_config(O, A, hid, act_limit=act_limit, TD_bound=td_bound, bound=over.get("bound", True),
                         policy_hidden=over.get("policy_hidden_sizes"))
    cfg["pad_to"] = getattr(alg.engine.layout, "pad_to", None)   # stored widths of the HIP arenas: the oracle's FLAT views follow them
    orc = DsacV1Oracle(cfg, state_dict={k: v.cpu() for k, v in alg.networks.state_dict().items()})
    return alg, orc


def run_case(title, O, A, hid, B, steps, act_limit=0.4, init=None, golden=None, **over):
    rep = Report(title)
    alg, orc = make_pair(O, A, hid, B, act_limit=act_limit, init=init, **over)
    e = alg.engine
    lay = e.layout
    assert lay.n_online == orc.flat_params().numel()
    rng = np.random.default_rng(9)
    cfg = orc.cfg
    noise_b = AdamNoise([("q", lay.n_q, cfg["lr_q"]), ("policy", lay.n_pi, cfg["lr_pi"]), ("log_alpha", 1, cfg["lr_alpha"])])
    tau = cfg["tau"]
    for it in range(steps):
        if golden is not None:
            data = {k: torch.as_tensor(golden["s%d/%s" % (it, k)]) for k in ("obs", "obs2", "act", "rew", "done")}
            noise = {k: torch.as_tensor(golden["s%d/%s" % (it, k)]) for k in ("eps_new", "eps_2", "z_t")}
        else:
            data = synth_batch(rng, B, O, A, lim=act_limit, p_done=0.05)
            torch.manual_seed(3000 + it)
            noise = draw_noise_v1(B, A)
        tb_ref = orc.compute_gradient(data, noise)
        e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
        e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z_t"].numpy(), noise["z_t"].numpy())
        e.compute_grads(it)
        e.sync()
        g, g_ref = e.grads.cpu().numpy(), orc.flat_grads().numpy()
        off = 0
        for net, n in (("q", lay.n_q), ("policy", lay.n_pi), ("log_alpha", 1)):
            rep.cmp("it%d grad.%s" % (it, net), g[off:off + n], g_ref[off:off + n], 1e-9, 3e-5)
            off += n
        delayed = it % cfg["delay_update"] == 0
        noise_b.step(g_ref, g, ("q",) + (("policy", "log_alpha") if delayed else ()))
        e.apply_update(it)
        orc.update(it)
        from dsac_v1_hip import LazyTbInfoV1
        alg._serial += 1
        tb = LazyTbInfoV1(alg, alg._serial, 0.0)
        assert list(tb.keys()) == V1_TB_KEYS
        for k in V1_TB_KEYS[:-1]:
            if k.startswith("Loss/Critic"):
                rep.cmp("it%d %s" % (it, k.split("/")[-1][:18]), [float(tb[k])], [float(tb_ref[k])], 1e-6, 1e-5)
            else:
                rep.cmp("it%d %s" % (it, k.split("/")[-1][:18]), [float(tb[k])], [float(tb_ref[k])], 1e-4)
        p_hip, t_hip, nb = e.online.cpu().numpy(), e.target.cpu().numpy(), noise_b.bound
        n_t = t_hip.size
        tb_t = tau * (it + 1)
        if golden is not None:
            crit = [i for i, k in enumerate(V1_TB_KEYS[:-1]) if k.startswith("Loss/Critic")]
            keep_i = [i for i in range(len(V1_TB_KEYS) - 1) if i not in crit]
            tb_g = np.asarray(golden["s%d/tb" % it], np.float64)
            rep.cmp("it%d tb vs reference" % it, [float(tb[V1_TB_KEYS[i]]) for i in keep_i], tb_g[keep_i], 1e-4)
            rep.cmp("it%d critic loss vs reference" % it, [float(tb[V1_TB_KEYS[i]]) for i in crit], tb_g[crit], 1e-6, 1e-5)
            rep.cmp_params("it%d params vs reference" % it, p_hip, golden["s%d/params" % it], nb, 1e-6, noise_b.lr_steps)
            rep.cmp_params("it%d targets vs reference" % it, t_hip, golden["s%d/targets" % it], tb_t * nb[:n_t], 1e-7,
                           tb_t * noise_b.lr_steps)
        rep.cmp_params("it%d params" % it, p_hip, orc.flat_params(), nb, 1e-6, noise_b.lr_steps)
        rep.cmp_params("it%d targets" % it, t_hip, orc.flat_targets(), tb_t * nb[:n_t], 1e-7, tb_t * noise_b.lr_steps)
    assert e.get_state()["adam_steps"][0] == steps
    rep.finish()


def test_v1_against_reference_golden():
    z = np.load(os.path.join(GOLDEN, "step_v1_tiny.npz"))
    init = {k[len("init/"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("init/")}
    run_case("v1 golden tiny", int(z["cfg_obs_dim"]), int(z["cfg_act_dim"]), tuple(int(h) for h in z["cfg_hidden"]),
             int(z["cfg_batch"]), int(z["cfg_steps"]), act_limit=float(z["cfg_act_limit"]), init=init, golden=z)


def test_v1_humanoid_shapes():
    run_case("v1 humanoid 3x256 B=256", 376, 17, (256, 256, 256), 256, steps=3)


def test_v1_runs_on_the_row_slice_chains(monkeypatch):
    """round 4: DSAC_V1 with MLP nets of equal width takes the row-slice chain kernels (one critic = fewer units in the same
    launches, the V1 row phase in the critics' backward chain): chain path == tile path within fp32 summation order on
    every intermediate the debug buffers expose, and both against the oracle (the cases above)."""
    a1, _ = make_pair(376, 17, (256, 256, 256), 256, seed=3)
    assert a1.engine.chain_active
    monkeypatch.setenv("DSACT_NO_CHAIN_V1", "1")
    a2, _ = make_pair(376, 17, (256, 256, 256), 256, seed=3)
    assert not a2.engine.chain_active
    rng = np.random.default_rng(4)
    for it in range(3):
        data = synth_batch(rng, 256, 376, 17)
        for a in (a1, a2):
            torch.manual_seed(90 + it)
            a.local_update(data, it)
    a1.engine.sync(); a2.engine.sync()
    for n in ("qout_c0", "qout_t0", "qout_p0", "logp_new", "d_new_act", "dZ.q1c.0", "dZ.pi.0", "H.q1p.2"):
        x, y = a1.engine.debug_read(n), a2.engine.debug_read(n)
        assert np.abs(x - y).max() <= 2e-5 * max(1.0, np.abs(y).max()), n
    d = (a1.engine.online - a2.engine.online).abs().max().item()
    assert d < 5e-6, d


def test_v1_unbounded_critic_loss():
    """`bound=False` (dsac_v1.py:227-228): the critic loss is -Normal(q, std).log_prob(target_q); also switchable on a live
    algorithm through `adjustable_parameters`."""
    run_case("v1 bound=False O=11 A=3 (64,64) B=64", 11, 3, (64, 64), 64, steps=3, bound=False)
    run_case("v1 bound=False humanoid 3x256 B=256", 376, 17, (256, 256, 256), 256, steps=2, bound=False)
    a1, _ = make_pair(11, 3, (64, 64), 64, seed=4)
    a2, _ = make_pair(11, 3, (64, 64), 64, seed=4, bound=False)
    assert "bound" in a1.adjustable_parameters and a1.bound is True and a2.bound is False
    a1.bound = False                       # reaches the engine like the reference's attribute is re-read every update
    rng = np.random.default_rng(1)
    for it in range(3):
        data = synth_batch(rng, 64, 11, 3)
        for a in (a1, a2):
            torch.manual_seed(70 + it)
            a.local_update(data, it)
    a1.engine.sync(); a2.engine.sync()
    assert torch.equal(a1.engine.online, a2.engine.online)


def test_v1_large_batch_tiles_and_split_k():
    run_case("v1 humanoid 3x256 B=512 (64x64 stage tiles, split-K dW)", 376, 17, (256, 256, 256), 512, steps=2)


def test_v1_ragged_and_one_dim_action():
    run_case("v1 ragged O=11 A=3 (96,40) B=50", 11, 3, (96, 40), 50, steps=3)
    run_case("v1 O=3 A=1 (64,64) B=64", 3, 1, (64, 64), 64, steps=3, act_limit=2.0)


def test_v1_ragged_and_unequal_widths_on_the_padded_chains():
    """round 6: ragged widths, and value_hidden_sizes != policy_hidden_sizes of the same depth, stored zero-padded (dsact/layout.py
    ArenaLayout pad_to) -- DSAC_V1 on the row-slice chains at shapes it ran on the tile stages (or refused) before"""
    for O, A, hid, B, over in ((24, 6, (96, 40), 64, {}), (17, 6, (200, 200, 200), 256, {}), (24, 6, (64, 64), 64, {"policy_hidden_sizes": [32, 48]}),
                               (11, 3, (128, 128), 128, {"policy_hidden_sizes": [256, 200]})):
        alg, _ = make_pair(O, A, hid, B, hip_pad_widths=True, **over)
        assert alg.engine.chain_active and alg.engine.layout.pad_to in (64, 128, 256)
        alg.engine.close()
        run_case("v1 padded O=%d A=%d %s B=%d %s" % (O, A, hid, B, over), O, A, hid, B, steps=3, hip_pad_widths=True, **over)
    from dsac_v1_hip import DSAC_V1_HIP
    with pytest.raises(NotImplementedError):     # another depth: no padded form; the tile-stage form of unequal lists is DSAC_V2_HIP's
        DSAC_V1_HIP(**hip_kwargs(24, 6, (64, 64), 64, algorithm="DSAC_V1_HIP", TD_bound=10.0, hip_pad_widths=True, policy_hidden_sizes=[64]))
    with pytest.raises(NotImplementedError):     # the exact layout was asked for
        DSAC_V1_HIP(**hip_kwargs(24, 6, (64, 64), 64, algorithm="DSAC_V1_HIP", TD_bound=10.0, hip_pad_widths=False, policy_hidden_sizes=[32, 48]))


def test_v1_local_update_surface_and_fused_equals_split():
    a1, orc = make_pair(11, 3, (64, 64), 64, seed=2)
    a2, _ = make_pair(11, 3, (64, 64), 64, seed=2)
    rng = np.random.default_rng(1)
    for it in range(4):
        data = synth_batch(rng, 64, 11, 3)
        torch.manual_seed(40 + it)
        noise = draw_noise_v1(64, 3)
        torch.manual_seed(40 + it)
        tb = a1.local_update(data, it)            # strict_rng: draws the reference's 5 tensors from the global RNG
        ref = orc.local_update(data, noise, it)
        assert list(tb.keys()) == V1_TB_KEYS
        for k in V1_TB_KEYS[:-1]:
            assert abs(float(tb[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (it, k)
        torch.manual_seed(40 + it)
        _, info = a2.get_remote_update_info(data, it)
        assert set(info) == {"q_grad", "policy_grad", "log_alpha_grad", "iteration"}
        a2.remote_update(info)
    s1, s2 = a1.networks.state_dict(), a2.networks.state_dict()
    assert list(s1.keys())[:2] == ["log_alpha", "q.q.0.weight"]
    for k in s1:
        assert torch.equal(s1[k].cpu(), s2[k].cpu()), k


@pytest.mark.parametrize("per_graph,total,O,tiles", [(2, 8, 17, True), (3, 6, 17, True), (4, 8, 16, False), (3, 9, 16, False),
                                                      (2, 8, 17, False), (3, 9, 11, False)])
def test_v1_graph_replay_equals_eager_steps(per_graph, total, O, tiles, monkeypatch):
    """DSAC_V1 through the graph flow == eager updates, bit for bit. tiles: the tile-stage kernels (DSACT_NO_CHAIN_V1; gather of
    the next update and the bookkeeping ride in k_loss_v1's launch, the single critic's first-layer tiles keep the padded copies
    fresh); else the row-slice chains, whose graph is the pipelined one (policy units of the next minibatch precomputed,
    discarded policy backward deferred, tagged hand-over) -- round 6: at any observation width (17, 11: the action columns of
    the critic's packed first layer are placed element-wise)."""
    A, hid, B, N = 4, (64, 64), 64, 2048
    if tiles:
        monkeypatch.setenv("DSACT_NO_CHAIN_V1", "1")
    engines = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hid, B, seed=4)
        e = alg.engine
        e.set_device_rng(777)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(8, B)))
        assert e.chain_active == (not tiles)
        if mode == "graph":
            e.graph_build(per_graph)
            assert e.debug_get("pipe_graph") == (0.0 if tiles else 1.0)
            e.graph_run(0, total)
        else:
            assert e.time_steps(0, total, use_graph=False) > 0
        e.sync()
        engines.append(e)
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(engines[0], name), getattr(engines[1], name)), name
    assert engines[0].get_state() == engines[1].get_state()
    assert torch.isfinite(engines[1].online).all()
    b0, b1 = engines[0].read_batch(with_logp=False), engines[1].read_batch(with_logp=False)
    for k in ("obs", "act", "rew", "obs2", "done"):
        assert np.array_equal(b0[k], b1[k]), k
