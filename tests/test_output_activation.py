"""value_output_activation / policy_output_activation other than "linear" (reference utils/common_utils.py:16-45 -> the module
behind the last Linear of networks/mlp.py:15-20; kwargs of SURVEY.md section 8 rows a10 / a12). Round 5 served them with the
tile-stage kernels (k_heads / k_loss / k_heads_bwd / k_policy_out apply the activation and its derivative, expressed through the
stored post-activation outputs); round 6 also on the row-slice chains (the generic-activation instantiations of the forward kernels
apply it in the heads, the backward row phases multiply by out_act_grad_y) and in both acting forwards (host / one launch)."""
import numpy as np
import pytest
import torch


def test_unsupported_output_activation_combinations_are_refused():
    from dsac_v2_hip import _check_supported
    import dsac_v1_hip

    kw = dict(obsv_dim=8, action_dim=2, value_hidden_sizes=[64, 64], policy_hidden_sizes=[64, 64], value_output_activation="tanh")
    _check_supported(kw)
    with pytest.raises(NotImplementedError):
        dsac_v1_hip._check_supported(kw)
    _check_supported(dict(kw, value_output_activation="gelu"))       # round 6: built (tile-stage kernels)
    with pytest.raises(NotImplementedError):
        _check_supported(dict(kw, value_output_activation="swish"))  # not one of the reference's names
    with pytest.raises(NotImplementedError):
        _check_supported(dict(kw, value_func_type="CNN", policy_func_type="CNN", value_conv_type="type_2", policy_conv_type="type_2",
                              obsv_dim=(3, 96, 96)))


def test_host_closed_forms_of_the_output_activations():
    """out_act_fwd / out_act_grad_y (dsact_math.h, compiled for the host) against torch's modules and autograd"""
    import ctypes
    import os

    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libdsact_hostmath.so"))
    lib.hm_out_act.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    mods = {1: torch.nn.ReLU(), 2: torch.nn.ELU(), 3: torch.nn.SELU(), 4: torch.nn.Sigmoid(), 5: torch.nn.Tanh(), 0: torch.nn.Identity(),
            6: torch.nn.GELU()}   # (6: the derivative is taken from the pre-activation, out_act_grad)
    zs = np.concatenate([np.linspace(-6, 6, 97), [0.0, 1e-3, -1e-3]]).astype(np.float32)
    for act, mod in mods.items():
        z = torch.tensor(zs, requires_grad=True)
        y = mod(z)
        y.sum().backward()
        for i, zv in enumerate(zs):
            yo, go = ctypes.c_float(), ctypes.c_float()
            lib.hm_out_act(act, float(zv), ctypes.byref(yo), ctypes.byref(go))
            assert abs(yo.value - float(y[i])) <= 2e-6 * max(1.0, abs(float(y[i]))), (act, zv)
            if not (act == 1 and zv == 0.0):
                assert abs(go.value - float(z.grad[i])) <= 3e-6, (act, zv, go.value, float(z.grad[i]))


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B,over", [
    (24, 6, (64, 64), 64, {"value_output_activation": "tanh", "policy_output_activation": "tanh"}),
    (376, 17, (256, 256, 256), 256, {"value_output_activation": "tanh"}),                 # the BASELINE shape (row-slice chains since round 6)
    (376, 17, (256, 256, 256), 256, {"value_output_activation": "relu", "policy_output_activation": "elu"}),
    (376, 17, (256, 256, 256), 1024, {"value_output_activation": "tanh", "policy_output_activation": "tanh"}),   # batch 1024: chains, not the (linear-only) throughput-regime kernels
    (11, 3, (96, 40), 50, {"value_output_activation": "sigmoid", "policy_output_activation": "elu"}),
    (24, 6, (64, 64), 64, {"value_output_activation": "selu", "policy_output_activation": "sigmoid"}),
    (24, 6, (64, 64), 64, {"policy_output_activation": "tanh", "policy_std_type": "parameter"}),   # log_std is not activated
    (24, 6, (64, 64), 64, {"value_output_activation": "tanh", "policy_act_distribution": "GaussDistribution"}),
    # "gelu" as an OUTPUT activation (round 6: the heads store d y / d z beside y; tile-stage kernels)
    (24, 6, (64, 64), 64, {"value_output_activation": "gelu", "policy_output_activation": "gelu"}),
    (376, 17, (256, 256, 256), 256, {"value_output_activation": "gelu"}),
    (11, 3, (96, 40), 50, {"policy_output_activation": "gelu", "value_output_activation": "tanh"}),
    (24, 6, (64, 64), 512, {"value_output_activation": "gelu", "policy_output_activation": "gelu", "policy_std_type": "parameter"}),
    (24, 6, (64, 64), 64, {"policy_output_activation": "gelu", "policy_std_type": "mlp_separated"}),
])
def test_output_activations_against_the_oracle(O, A, hid, B, over):
    """every intermediate, gradient, statistic and parameter against the oracle, which is pinned bit-exact to the live
    reference with these kwargs (tests/test_oracle_vs_reference.py::test_output_activations_bit_exact_vs_live_reference)"""
    from test_hip_parity import run_case

    run_case("output activations O=%d A=%d hid=%s B=%d %s" % (O, A, hid, B, over), O, A, hid, B, steps=3, **over)


@pytest.mark.gpu
def test_output_activation_runs_on_the_chains_and_in_both_acting_forwards(monkeypatch):
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair

    O, A, hid, B = 24, 6, (64, 64), 64
    alg, orc = make_pair(O, A, hid, B, seed=2, value_output_activation="tanh", policy_output_activation="tanh")
    e = alg.engine
    assert e.chain_active and e.debug_get("act_fast") == 1.0        # round 5: tile stages + the general acting path
    monkeypatch.setenv("DSACT_NO_CHAIN", "1")
    tiles, _ = make_pair(O, A, hid, B, seed=2, value_output_activation="tanh", policy_output_activation="tanh")
    monkeypatch.delenv("DSACT_NO_CHAIN")
    assert not tiles.engine.chain_active
    obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
    want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
    assert np.all(np.abs(want[:, :A]) <= 1.0)        # tanh-activated means
    for eng in (e, tiles.engine):
        for host in (1, 0):                          # acting forward on the host / as one launch; 3 rows at once: the tile-stage forward
            eng.debug_set("host_act", host)
            np.testing.assert_allclose(np.concatenate([eng.policy_forward(obs[i:i + 1]) for i in range(3)]), want, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(eng.policy_forward(obs), want, atol=2e-5, rtol=1e-5)
    e.debug_set("host_act", 1)
    # the attached module's forward (what samplers / evaluators call) goes through the same kernels
    got = alg.networks.policy(torch.as_tensor(obs)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    # the sampling step with the activation in front of the tanh-Gaussian: host == one launch
    eps = np.full((1, A), 0.25, np.float32)
    outs = []
    for host in (1, 0):
        e.debug_set("host_act", host)
        a, lp = e.act_sample(obs[0], eps)
        outs.append((a.copy(), float(lp[0])))
    e.debug_set("host_act", 1)
    np.testing.assert_allclose(outs[0][0], outs[1][0], atol=2e-6, rtol=0)
    assert abs(outs[0][1] - outs[1][1]) <= 5e-4


@pytest.mark.gpu
def test_output_activation_graph_replays_equal_eager_updates():
    from test_hip_parity import make_pair

    O, A, hid, B, N = 16, 4, (64, 64), 64, 2048
    algs = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hid, B, seed=4, value_output_activation="tanh", policy_output_activation="sigmoid")
        e = alg.engine
        e.set_device_rng(777)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(7, B)))
        if mode == "graph":
            e.graph_build(4)
            e.graph_run(1, 12)
        else:
            assert e.time_steps(1, 12, use_graph=False) > 0
        e.sync()
        algs.append(alg)
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(algs[0].engine, name), getattr(algs[1].engine, name)), name
    assert torch.isfinite(algs[1].engine.online).all()
