"""policy_std_type = "parameter" (reference networks/mlp.py:63-73,92-97; a kwarg of SURVEY.md section 8 rows a10 / a11): the
MLP gives the mean, log_std is a learnable (1, act_dim) parameter. The HIP arenas keep the (2 act_dim x H) output layer of
"mlp_shared": rows [act_dim, 2 act_dim) of its weight are structurally zero (gradient masked), the second half of its bias IS
log_std (dsac-v2_amd/dsact/layout.py, include/dsact.h policy_std_param)."""
import numpy as np
import pytest
import torch

from dsact.layout import ArenaLayout
from oracle import ref_loader
from oracle.dsact_oracle import DsactOracle, default_config


def test_layout_of_the_parameter_std_type():
    O, A, hid = 24, 6, [64, 64]
    lay = ArenaLayout(O, A, hid, policy_std_type="parameter")
    same = ArenaLayout(O, A, hid)
    assert (lay.n_pi, lay.n_online, lay.n_target) == (same.n_pi, same.n_online, same.n_target)   # the arenas do not change shape
    sl = lay.param_slices("policy")
    names = [s[0] for s in sl]
    assert names == ["log_std", "mean.0.weight", "mean.0.bias", "mean.2.weight", "mean.2.bias", "mean.4.weight", "mean.4.bias"]
    by = {s[0]: s for s in sl}
    w_out = dict((s[0], s) for s in same.param_slices("policy"))["policy.4.weight"]
    b_out = dict((s[0], s) for s in same.param_slices("policy"))["policy.4.bias"]
    assert by["mean.4.weight"][2] == w_out[2] and by["mean.4.weight"][3] == (A, 64)           # the first A rows of the shared layer
    assert by["mean.4.bias"][2] == b_out[2] and by["mean.4.bias"][3] == (A,)
    assert by["log_std"][2] == b_out[2] + A and by["log_std"][3] == (1, A)                     # the second half of its bias
    arena, off, cnt = lay.zero_rows("policy")
    assert (arena, off, cnt) == ("online", w_out[2] + A * 64, A * 64) and lay.zero_rows("q1") is None and same.zero_rows("policy") is None
    assert lay.zero_rows("policy_target")[0] == "target"
    # state_dict keys in the order of the oracle (== the reference's, tests/test_oracle_vs_reference.py)
    torch.manual_seed(0)
    orc = DsactOracle(default_config(O, A, hid, policy_std_type="parameter"))
    assert list(lay.state_dict_keys().keys()) == list(orc.state_dict().keys())
    assert all(tuple(v.shape) == tuple(lay.state_dict_keys()[k]) for k, v in orc.state_dict().items())


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")
def test_container_with_parameter_std_matches_the_reference_module():
    """same seed -> same initial state_dict (keys, order, values); forward of the stand-alone CPU module == reference's"""
    from dsac_v2_hip import ApproxContainer

    ref = ref_loader.import_reference()
    O, A, hid = 24, 6, (64, 64)
    kw = ref_loader.reference_kwargs(O, A, hid, policy_std_type="parameter")
    torch.manual_seed(3)
    theirs = ref.ApproxContainer(**kw)
    torch.manual_seed(3)
    ours = ApproxContainer(**kw)
    sd, osd = theirs.state_dict(), ours.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    assert [n for n, _ in theirs.policy.named_parameters()] == [n for n, _ in ours.policy.named_parameters()]
    with torch.no_grad():
        ours.policy.log_std.add_(torch.linspace(-1, 1, A)[None])
    theirs.load_state_dict(ours.state_dict())
    obs = torch.randn(5, O)
    assert torch.equal(theirs.policy(obs), ours.policy(obs))


def test_unsupported_combinations_are_refused():
    from dsac_v2_hip import _check_supported
    import dsac_v1_hip

    kw = dict(obsv_dim=8, action_dim=2, value_hidden_sizes=[64, 64], policy_hidden_sizes=[64, 64], policy_std_type="no_such_type")
    with pytest.raises(NotImplementedError):
        _check_supported(kw)
    kw["policy_std_type"] = "parameter"
    _check_supported(kw)
    with pytest.raises(NotImplementedError):
        dsac_v1_hip._check_supported(kw)
    kw.update(value_func_type="CNN", policy_func_type="CNN", value_conv_type="type_2", policy_conv_type="type_2", obsv_dim=(3, 96, 96))
    with pytest.raises(NotImplementedError):
        _check_supported(kw)


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B,over", [
    (24, 6, (64, 64), 64, {}),                                     # chain path
    (376, 17, (256, 256, 256), 256, {}),                           # the BASELINE shape
    (11, 3, (96, 40), 50, {}),                                     # tile path (ragged widths, odd observation width)
    (376, 17, (256, 256, 256), 1024, {}),                          # throughput-regime kernels
    (23, 5, (96, 40), 512, {}),                                    # tile path with split-K weight gradients (batch > 448): ADVICE r5
    (24, 6, (64, 64), 64, {"policy_act_distribution": "GaussDistribution"}),
])
def test_parameter_std_against_the_oracle(O, A, hid, B, over):
    """every intermediate, gradient (log_std's = the bias-tail rows), statistic and parameter against the oracle, which is
    pinned bit-exact to the live reference with this kwarg (tests/test_oracle_vs_reference.py)"""
    from test_hip_parity import run_case

    run_case("std parameter O=%d A=%d hid=%s B=%d %s" % (O, A, hid, B, over), O, A, hid, B, steps=3, policy_std_type="parameter", **over)


@pytest.mark.gpu
def test_parameter_std_structure_survives_graph_replays_and_the_plugin_surface():
    """the structurally-zero rows stay exactly zero through pipelined graph replays (online, target and both Adam moments),
    log_std moves, graph == eager bitwise, and get_remote_update_info hands the gradients out in the reference's parameter
    order (log_std first, shape (1, act_dim))"""
    from test_hip_parity import make_pair
    from helpers import synth_batch

    O, A, hid, B, N = 16, 4, (64, 64), 64, 2048
    algs = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hid, B, seed=4, policy_std_type="parameter")
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
            assert e.debug_get("pipe_graph") == 1.0
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
        arena, off, cnt = lay.zero_rows(net)
        assert not getattr(e1, arena)[off:off + cnt].any(), net
    _, off, cnt = lay.zero_rows("policy")
    assert not e1.adam_m[off:off + cnt].any() and not e1.adam_v[off:off + cnt].any()
    pol = algs[1].networks.policy
    assert tuple(pol.log_std.shape) == (1, A) and not torch.equal(pol.log_std.cpu(), torch.full((1, A), -0.5))
    assert torch.isfinite(e1.online).all()
    # the plugin surface: gradients in the reference's parameter order
    data = synth_batch(np.random.default_rng(0), B, O, A)
    torch.manual_seed(5)
    _, info = algs[1].get_remote_update_info({k: v for k, v in data.items()}, 13)
    shapes = [tuple(t.shape) for t in info["policy_grad"]]
    assert shapes == [tuple(p.shape) for p in pol.parameters()] and shapes[0] == (1, A)
    assert info["policy_grad"][0].abs().sum() > 0
    before = pol.log_std.detach().clone()
    algs[1].remote_update(info)
    algs[1].engine.sync()
    assert not getattr(e1, "online")[off:off + cnt].any()
    assert algs[1].networks.state_dict()["policy.log_std"].shape == (1, A)
    del before
