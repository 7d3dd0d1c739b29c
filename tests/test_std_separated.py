"""policy_std_type = "mlp_separated" (reference networks/mlp.py:46-57,80-85; a kwarg of SURVEY.md section 8 rows a10 / a11): the
policy is TWO MLPs over the observation, `mean` and `log_std`. The HIP arenas keep them side by side exactly like the CNN nets'
twin trunks (dsac-v2_amd/dsact/layout.py twin_mlp_views, include/dsact.h policy_twin): layer 0 one dense [mean ; log_std] matrix,
hidden layers two blocks, the output layer one (2 act_dim x 2H) matrix whose off-diagonal blocks are structurally zero."""
import numpy as np
import pytest
import torch

from dsact.layout import ArenaLayout
from oracle import ref_loader
from oracle.dsact_oracle import DsactOracle, default_config


def test_layout_of_the_separated_std_type():
    O, A, hid = 24, 6, [64, 48]
    lay = ArenaLayout(O, A, hid, policy_std_type="mlp_separated")
    one = sum(o * i + o for o, i in ((64, O), (48, 64), (A, 48)))
    assert lay.n_pi == 2 * one + 2 * A * 48                              # two trunks + the output layer's two zero blocks
    names = [v[0] for v in lay.param_views("policy")]
    assert names == ["mean.0.weight", "mean.0.bias", "mean.2.weight", "mean.2.bias", "mean.4.weight", "mean.4.bias",
                     "log_std.0.weight", "log_std.0.bias", "log_std.2.weight", "log_std.2.bias", "log_std.4.weight", "log_std.4.bias"]
    by = {v[0]: v for v in lay.param_views("policy")}
    base = lay.net_offset["policy"][1]
    assert by["mean.0.weight"][2:] == (base, (64, O), (O, 1)) and by["log_std.0.weight"][2] == base + 64 * O     # [mean ; log_std]
    assert by["mean.4.weight"][3:] == ((A, 48), (96, 1)) and by["log_std.4.weight"][2] == by["mean.4.weight"][2] + A * 96 + 48
    # every float of the policy region is either a parameter or inside a zero block, exactly once
    cover = np.zeros(lay.n_pi, np.int32)
    flat = torch.arange(lay.n_online)
    for _, _, off, shape, strides in lay.param_views("policy"):
        cover[torch.as_strided(flat, shape, strides, off).reshape(-1).numpy() - base] += 1
    for _, off, shape, strides in lay.zero_blocks("policy"):
        cover[torch.as_strided(flat, shape, strides, off).reshape(-1).numpy() - base] += 1
    assert (cover == 1).all()
    assert lay.zero_blocks("q1") == [] and ArenaLayout(O, A, hid).zero_blocks("policy") == []
    assert lay.zero_blocks("policy_target")[0][0] == "target"
    # state_dict keys in the order of the oracle (== the reference's, tests/test_oracle_vs_reference.py)
    torch.manual_seed(0)
    orc = DsactOracle(default_config(O, A, hid, policy_std_type="mlp_separated"))
    assert list(lay.state_dict_keys().keys()) == list(orc.state_dict().keys())
    assert all(tuple(v.shape) == tuple(lay.state_dict_keys()[k]) for k, v in orc.state_dict().items())
    # the oracle's arena-order flat view has the layout's size and zeros exactly in the zero blocks
    fp = orc.flat_params()
    assert fp.numel() == lay.n_online
    for _, off, shape, strides in lay.zero_blocks("policy"):
        assert not torch.as_strided(fp, shape, strides, off).any()
    for name, _, off, shape, strides in lay.param_views("policy"):
        assert torch.equal(torch.as_strided(fp, shape, strides, off), orc.state_dict()["policy." + name]), name


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")
def test_container_with_separated_std_matches_the_reference_module():
    """same seed -> same initial state_dict (keys, order, values); forward of the stand-alone CPU module == reference's"""
    from dsac_v2_hip import ApproxContainer

    ref = ref_loader.import_reference()
    O, A, hid = 24, 6, (64, 64)
    kw = ref_loader.reference_kwargs(O, A, hid, policy_std_type="mlp_separated")
    torch.manual_seed(3)
    theirs = ref.ApproxContainer(**kw)
    torch.manual_seed(3)
    ours = ApproxContainer(**kw)
    sd, osd = theirs.state_dict(), ours.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    assert [n for n, _ in theirs.policy.named_parameters()] == [n for n, _ in ours.policy.named_parameters()]
    obs = torch.randn(5, O)
    assert torch.equal(theirs.policy(obs), ours.policy(obs))


def test_unsupported_combinations_are_refused():
    from dsac_v2_hip import _check_supported
    import dsac_v1_hip

    kw = dict(obsv_dim=8, action_dim=2, value_hidden_sizes=[64, 64], policy_hidden_sizes=[64, 64], policy_std_type="mlp_separated")
    _check_supported(kw)
    with pytest.raises(NotImplementedError):
        dsac_v1_hip._check_supported(kw)
    kw.update(value_func_type="CNN", policy_func_type="CNN", value_conv_type="type_2", policy_conv_type="type_2", obsv_dim=(3, 96, 96))
    with pytest.raises(NotImplementedError):
        _check_supported(kw)


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B,over", [
    (24, 6, (64, 64), 64, {}),
    (376, 17, (256, 256, 256), 256, {}),                           # the BASELINE shape: policy rows 512 wide beside 256-wide critics
    (11, 3, (96, 40), 50, {}),                                     # ragged widths, odd batch
    (24, 6, (64, 64), 64, {"policy_hidden_sizes": [32, 48]}),      # with value_hidden_sizes != policy_hidden_sizes
    (23, 5, (96, 40), 512, {}),                                    # split-K weight gradients (batch > 448)
    (24, 6, (64, 64), 64, {"policy_hidden_activation": "relu", "policy_output_activation": "tanh"}),
    (24, 6, (64, 64), 64, {"policy_act_distribution": "GaussDistribution"}),
    (16, 4, (64,), 64, {}),                                        # one hidden layer: no block layer at all
])
def test_separated_std_against_the_oracle(O, A, hid, B, over):
    """every intermediate (rows [z_mean | z_log_std]), gradient, statistic and parameter against the oracle, which is pinned bit-exact
    to the live reference with this kwarg (tests/test_oracle_vs_reference.py::test_std_type_mlp_separated_bit_exact_vs_live_reference)"""
    from test_hip_parity import run_case

    run_case("std mlp_separated O=%d A=%d hid=%s B=%d %s" % (O, A, hid, B, over), O, A, hid, B, steps=3, policy_std_type="mlp_separated", **over)


@pytest.mark.gpu
def test_separated_std_acting_forwards_agree_with_the_oracle():
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair

    for O, A, hid in ((24, 6, (64, 64)), (376, 17, (256, 256, 256)), (11, 3, (96, 40, 64))):
        alg, orc = make_pair(O, A, hid, 64, seed=2, policy_std_type="mlp_separated")
        e = alg.engine
        assert not e.chain_active and e.debug_get("act_fast") == 1.0
        obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
        want = policy_forward(torch.as_tensor(obs), [t.detach() for t in orc.p["policy"]], orc.cfg).numpy()
        np.testing.assert_allclose(e.policy_forward(obs), want, atol=2e-5, rtol=1e-5)                      # tile stages (3 rows)
        e.debug_set("host_act", 1)
        host = np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(3)])
        assert e.debug_get("act_host") == 1.0
        e.debug_set("host_act", 0)
        one = np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(3)])                           # one-launch GPU forward
        assert e.debug_get("act_host") == 0.0
        np.testing.assert_allclose(host, want, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(one, want, atol=2e-5, rtol=1e-5)
        # dsact_act_sample == TanhGaussDistribution.sample() on the same logits and generator state, on both acting forwards
        for mode in (1, 0):
            e.debug_set("host_act", mode)
            for i in range(4):
                torch.manual_seed(i)
                eps = torch.randn(1, A)
                action, logp = e.act_sample(obs[0], eps.numpy())
                dist = alg.networks.create_action_distributions(torch.from_numpy(e.policy_forward(obs[:1])))
                torch.manual_seed(i)
                a_ref, lp_ref = dist.sample()
                np.testing.assert_allclose(action, a_ref[0].numpy(), atol=2e-6, rtol=0)
                assert abs(float(logp[0]) - float(lp_ref[0])) <= 2e-4


@pytest.mark.gpu
def test_separated_std_structure_survives_graph_replays_and_checkpoints(tmp_path):
    """graph == eager bitwise; the output layer's zero blocks stay exactly zero in online, target and both Adam moments; the state_dict
    has the reference's keys and loads into a fresh container / the oracle; remote_update takes gradients in parameter order"""
    from test_hip_parity import make_pair
    from helpers import synth_batch

    O, A, hid, B, N = 16, 4, (64, 64), 64, 2048
    algs = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hid, B, seed=4, policy_std_type="mlp_separated")
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
    e0, e1 = algs[0].engine, algs[1].engine
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(e0, name), getattr(e1, name)), name
    lay = e1.layout
    for net in ("policy", "policy_target"):
        for arena, off, shape, strides in lay.zero_blocks(net):
            assert not torch.as_strided(getattr(e1, arena), shape, strides, off).any(), net
    for _, off, shape, strides in lay.zero_blocks("policy"):
        assert not torch.as_strided(e1.adam_m, shape, strides, off).any() and not torch.as_strided(e1.adam_v, shape, strides, off).any()
    assert torch.isfinite(e1.online).all()
    pol = algs[1].networks.policy
    sd = algs[1].networks.state_dict()
    assert list(sd.keys()) == list(lay.state_dict_keys().keys())
    assert sd["policy.log_std.4.weight"].shape == (A, 64) and sd["policy.mean.2.weight"].shape == (64, 64)
    # the trained twin policy as a stand-alone CPU module (what samplers / evaluators load): same forward as the engine's
    from dsac_v2_hip import ApproxContainer
    from test_hip_parity import hip_kwargs
    cpu = ApproxContainer(**hip_kwargs(O, A, hid, B, policy_std_type="mlp_separated"))
    torch.save(sd, tmp_path / "apprfunc.pkl")
    cpu.load_state_dict(torch.load(tmp_path / "apprfunc.pkl", map_location="cpu"))
    obs = torch.randn(5, O)
    np.testing.assert_allclose(pol(obs).numpy(), cpu.policy(obs).detach().numpy(), atol=2e-5, rtol=1e-5)
    # the plugin surface: gradients in the reference's parameter order (mean.* then log_std.*)
    data = synth_batch(np.random.default_rng(0), B, O, A)
    torch.manual_seed(5)
    _, info = algs[1].get_remote_update_info({k: v for k, v in data.items()}, 13)
    assert [tuple(t.shape) for t in info["policy_grad"]] == [tuple(p_.shape) for p_ in pol.parameters()]
    assert all(float(t.abs().sum()) > 0 for t in info["policy_grad"])
    algs[1].remote_update(info)
    algs[1].engine.sync()
    for arena, off, shape, strides in lay.zero_blocks("policy"):
        assert not torch.as_strided(getattr(e1, arena), shape, strides, off).any()
