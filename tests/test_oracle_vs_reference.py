"""Pins oracle/dsact_oracle.py against the LIVE unmodified reference (only where /root/reference
is mounted, i.e. in the build container; skipped on the GPU box)."""
import numpy as np
import pytest
import torch

from oracle import ref_loader
from oracle.dsact_oracle import TB_KEYS, DsactOracle, default_config, draw_noise
from helpers import synth_batch

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")


@pytest.mark.parametrize("O,A,hid,B,va,pa,dist", [
    (376, 17, (256, 256, 256), 256, "gelu", "gelu", "TanhGaussDistribution"), (376, 17, (256, 256), 128, "gelu", "gelu", "TanhGaussDistribution"),
    (3, 1, (64, 64, 64), 64, "gelu", "gelu", "TanhGaussDistribution"),
    # value_hidden_activation / policy_hidden_activation other than the examples' gelu (utils/common_utils.py:16-45)
    (24, 6, (64, 64), 64, "relu", "tanh", "TanhGaussDistribution"), (24, 6, (64, 64), 64, "elu", "selu", "TanhGaussDistribution"),
    (24, 6, (64, 64), 64, "sigmoid", "relu", "TanhGaussDistribution"),
    # policy_act_distribution = GaussDistribution (utils/act_distribution_cls.py:82-115): no tanh squashing (round 5)
    (24, 6, (64, 64), 64, "gelu", "gelu", "GaussDistribution"), (376, 17, (256, 256, 256), 64, "gelu", "gelu", "GaussDistribution")])
def test_bit_exact_vs_live_reference(O, A, hid, B, va, pa, dist):
    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    kw = ref_loader.reference_kwargs(O, A, hid, value_hidden_activation=va, policy_hidden_activation=pa, policy_act_distribution=dist)
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = default_config(O, A, hid, value_act=va, policy_act=pa, act_dist=dist)
    torch.manual_seed(0)
    same_seed = DsactOracle(cfg)  # same construction order => same init from the same seed
    sd = alg.networks.state_dict()
    osd = same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactOracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets = alg.networks
        gref = torch.cat([p.grad.reshape(-1) for n in ("q1", "q2", "policy") for p in getattr(nets, n).parameters()]
                         + [nets.log_alpha.grad.reshape(1)])
        assert torch.equal(gref, orc.flat_grads())
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


@pytest.mark.parametrize("O,A,hid,B,dist", [(24, 6, (64, 64), 64, "TanhGaussDistribution"), (376, 17, (256, 256, 256), 64, "TanhGaussDistribution"),
                                            (11, 3, (64, 64), 32, "GaussDistribution")])
def test_std_type_parameter_bit_exact_vs_live_reference(O, A, hid, B, dist):
    """policy_std_type = "parameter" (networks/mlp.py:63-73,92-97: the MLP gives the mean, log_std is a learnable (1, act_dim)
    parameter): same seed -> same initial state, then four updates with every gradient and parameter equal under the
    reference's own parameter names (the oracle's flat views are in the HIP arena's padded order)."""
    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    kw = ref_loader.reference_kwargs(O, A, hid, policy_act_distribution=dist, policy_std_type="parameter")
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = default_config(O, A, hid, act_dist=dist, policy_std_type="parameter")
    torch.manual_seed(0)
    same_seed = DsactOracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactOracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets, og = alg.networks, orc.grad_dict()
        for n in ("q1", "q2", "policy"):
            for name, p_ in getattr(nets, n).named_parameters():
                assert torch.equal(p_.grad, og[n + "." + name]), (n, name)
        assert torch.equal(nets.log_alpha.grad, og["log_alpha"])
        # the padded flat view: the structurally-zero rows carry zeros, log_std's gradient sits in the bias tail
        fg, nq = orc.flat_grads(), sum(p_.numel() for p_ in nets.q1.parameters())
        H = hid[-1]
        tail = fg[2 * nq:-1][-(2 * A * H + 2 * A):]
        assert torch.equal(tail[A * H:2 * A * H], torch.zeros(A * H)) and torch.equal(tail[-A:], nets.policy.log_std.grad.reshape(-1))
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


@pytest.mark.parametrize("O,A,hid,B,vo,po,st", [(24, 6, (64, 64), 64, "tanh", "tanh", "mlp_shared"), (376, 17, (256, 256, 256), 64, "tanh", "linear", "mlp_shared"),
                                                (11, 3, (64, 64), 32, "sigmoid", "elu", "mlp_shared"), (24, 6, (64, 64), 64, "relu", "selu", "mlp_shared"),
                                                (24, 6, (64, 64), 64, "linear", "tanh", "parameter"),
                                                (24, 6, (64, 64), 64, "gelu", "gelu", "mlp_shared"), (11, 3, (96, 40), 50, "gelu", "tanh", "parameter")])
def test_output_activations_bit_exact_vs_live_reference(O, A, hid, B, vo, po, st):
    """value_output_activation / policy_output_activation other than "linear" (utils/common_utils.py:16-45 -> the module behind
    the last Linear, networks/mlp.py:15-20), alone and with policy_std_type "parameter" (whose log_std is NOT activated)."""
    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    kw = ref_loader.reference_kwargs(O, A, hid, value_output_activation=vo, policy_output_activation=po, policy_std_type=st)
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = default_config(O, A, hid, value_out_act=vo, policy_out_act=po, policy_std_type=st)
    torch.manual_seed(0)
    same_seed = DsactOracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactOracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets, og = alg.networks, orc.grad_dict()
        for n in ("q1", "q2", "policy"):
            for name, p_ in getattr(nets, n).named_parameters():
                assert torch.equal(p_.grad, og[n + "." + name]), (n, name)
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


@pytest.mark.parametrize("O,A,hid,hp,B,pa,po,dist", [
    (24, 6, (64, 64), None, 64, "gelu", "linear", "TanhGaussDistribution"), (376, 17, (256, 256, 256), None, 64, "gelu", "linear", "TanhGaussDistribution"),
    (11, 3, (96, 40), (40, 24), 32, "relu", "linear", "TanhGaussDistribution"), (24, 6, (64, 64), None, 64, "tanh", "tanh", "GaussDistribution")])
def test_std_type_mlp_separated_bit_exact_vs_live_reference(O, A, hid, hp, B, pa, po, dist):
    """policy_std_type = "mlp_separated" (networks/mlp.py:46-57,80-85: `mean` and `log_std` from two MLPs over the observation):
    same seed -> same initial state under the reference's names and order, then four updates with every gradient and parameter
    equal; the oracle's flat views are in the HIP arena's twin-trunk order (its structural zero blocks carry zeros)."""
    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    over = dict(policy_std_type="mlp_separated", policy_hidden_activation=pa, policy_output_activation=po, policy_act_distribution=dist)
    if hp is not None:
        over["policy_hidden_sizes"] = list(hp)
    kw = ref_loader.reference_kwargs(O, A, hid, **over)
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = default_config(O, A, hid, policy_std_type="mlp_separated", policy_act=pa, policy_out_act=po, act_dist=dist,
                         policy_hidden=list(hp) if hp is not None else None)
    torch.manual_seed(0)
    same_seed = DsactOracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactOracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it, keep=(it == 1))     # (keep: the [z_mean | z_log_std] collection changes nothing)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets, og = alg.networks, orc.grad_dict()
        for n in ("q1", "q2", "policy"):
            for name, p_ in getattr(nets, n).named_parameters():
                assert torch.equal(p_.grad, og[n + "." + name]), (n, name)
        assert torch.equal(nets.log_alpha.grad, og["log_alpha"])
        # the arena-order flat view: the output layer is the dense (2A x 2H) matrix [[w_mean, 0], [0, w_log_std]]
        fg, nq = orc.flat_grads(), sum(p_.numel() for p_ in nets.q1.parameters())
        H = (hp or hid)[-1]
        wout = fg[2 * nq:-1][-(4 * A * H + 2 * A):-(2 * A)].reshape(2 * A, 2 * H)
        assert torch.equal(wout[:A, :H], nets.policy.mean[-2].weight.grad) and torch.equal(wout[A:, H:], nets.policy.log_std[-2].weight.grad)
        assert not wout[:A, H:].any() and not wout[A:, :H].any()
        assert fg.numel() == 2 * nq + 2 * sum(p_.numel() for p_ in nets.policy.mean.parameters()) + 2 * A * H + 1
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


@pytest.mark.parametrize("O,A,hv,hp,B", [(24, 6, (64, 64), (32, 48), 64), (376, 17, (256, 256, 256), (128, 128, 128), 64), (11, 3, (96, 40), (40, 96), 32),
                                         # lists of different length (round 6)
                                         (24, 6, (64, 64, 64), (64, 64), 64), (24, 6, (64, 64), (48, 96, 32), 64), (376, 17, (256, 256, 256), (256, 256), 64),
                                         (11, 3, (40,), (96, 40, 24, 56), 32)])
def test_unequal_hidden_sizes_bit_exact_vs_live_reference(O, A, hv, hp, B):
    """value_hidden_sizes != policy_hidden_sizes (utils/common_utils.py:59-62 reads them per key): other widths, other depth"""
    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    kw = ref_loader.reference_kwargs(O, A, hv, policy_hidden_sizes=list(hp))
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = default_config(O, A, hv, policy_hidden=list(hp))
    torch.manual_seed(0)
    same_seed = DsactOracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactOracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets = alg.networks
        gref = torch.cat([p.grad.reshape(-1) for n in ("q1", "q2", "policy") for p in getattr(nets, n).parameters()]
                         + [nets.log_alpha.grad.reshape(1)])
        assert torch.equal(gref, orc.flat_grads())
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


def _cnn_kwargs(obs_shape, A, conv_type):
    kw = ref_loader.reference_kwargs(obs_shape, A, (256, 256, 256), act_limit=1.0)
    for key in ("value", "policy"):
        kw[key + "_func_type"] = "CNN"
        kw[key + "_conv_type"] = conv_type
        kw.pop(key + "_hidden_sizes")
    return kw


@pytest.mark.parametrize("obs_shape,A,conv_type,B", [((3, 96, 96), 3, "type_2", 8), ((4, 84, 84), 2, "type_1", 4)])
def test_cnn_bit_exact_vs_live_reference(obs_shape, A, conv_type, B):
    """the CNN approximators (networks/cnn.py) through the same update."""
    from oracle.dsact_oracle_cnn import DsactCnnOracle, cnn_config, synth_image_batch

    torch.set_num_threads(2)
    ref = ref_loader.import_reference()
    kw = _cnn_kwargs(obs_shape, A, conv_type)
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    cfg = cnn_config(obs_shape, A, conv_type)
    torch.manual_seed(0)
    same_seed = DsactCnnOracle(cfg)
    sd = alg.networks.state_dict()
    osd = same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsactCnnOracle(cfg, state_dict=sd)
    for it in range(3):
        d = synth_image_batch(cfg, B, seed=it)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        for k in TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets = alg.networks
        gref = torch.cat([p.grad.reshape(-1) for n in ("q1", "q2", "policy") for p in getattr(nets, n).parameters()]
                         + [nets.log_alpha.grad.reshape(1)])
        assert torch.equal(gref, orc.flat_grads())
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


@pytest.mark.parametrize("O,A,hid,B,bound,hp", [(11, 3, (64, 64), 32, True, None), (376, 17, (256, 256, 256), 64, True, None), (3, 1, (32, 32), 16, True, None),
                                                (11, 3, (64, 64), 32, False, None),
                                                # value_hidden_sizes != policy_hidden_sizes (round 6: DSAC_V1_HIP stores them zero-padded)
                                                (24, 6, (64, 64), 64, True, (32, 48)), (11, 3, (128, 128), 32, True, (256, 200))])
def test_v1_bit_exact_vs_live_reference(O, A, hid, B, bound, hp):
    """DSAC_V1 (reference dsac_v1.py) restated in oracle/dsac_v1_oracle.py -- both critic
    losses (`bound` True: dsac_v1.py:217-226, False: :227-228)."""
    import importlib

    from oracle.dsac_v1_oracle import V1_TB_KEYS, DsacV1Oracle, draw_noise_v1

    torch.set_num_threads(2)
    ref_loader.import_reference()
    v1 = importlib.import_module("dsac_v1")
    kw = ref_loader.reference_kwargs(O, A, hid, algorithm="DSAC_V1", TD_bound=10, bound=bound, **({"policy_hidden_sizes": list(hp)} if hp else {}))
    torch.manual_seed(0)
    alg = v1.DSAC_V1(**kw)
    cfg = default_config(O, A, hid, TD_bound=10, bound=bound, policy_hidden=list(hp) if hp else None)
    torch.manual_seed(0)
    same_seed = DsacV1Oracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsacV1Oracle(cfg, state_dict=sd)
    rng = np.random.default_rng(0)
    for it in range(4):
        d = synth_batch(rng, B, O, A)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise_v1(B, A), it)
        for k in V1_TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets = alg.networks
        gref = torch.cat([p.grad.reshape(-1) for n in ("q", "policy") for p in getattr(nets, n).parameters()]
                         + [nets.log_alpha.grad.reshape(1)])
        assert torch.equal(gref, orc.flat_grads())
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)


def test_v1_cnn_bit_exact_vs_live_reference():
    """VERDICT r3 missing #2: DSAC_V1 with the CNN approximators (example_train/dsacv1_cnn_carracing_offasync.py) --
    oracle/dsac_v1_oracle_cnn.py against the unmodified dsac_v1.py over networks/cnn.py, max-abs difference 0.0."""
    import importlib

    from oracle.dsac_v1_oracle import V1_TB_KEYS, draw_noise_v1
    from oracle.dsac_v1_oracle_cnn import DsacV1CnnOracle
    from oracle.dsact_oracle_cnn import cnn_config, synth_image_batch

    torch.set_num_threads(2)
    ref_loader.import_reference()
    v1 = importlib.import_module("dsac_v1")
    obs_shape, A, conv_type, B = (3, 96, 96), 3, "type_2", 4
    kw = dict(_cnn_kwargs(obs_shape, A, conv_type), algorithm="DSAC_V1", TD_bound=10, bound=True)
    torch.manual_seed(0)
    alg = v1.DSAC_V1(**kw)
    cfg = cnn_config(obs_shape, A, conv_type, TD_bound=10, bound=True)
    torch.manual_seed(0)
    same_seed = DsacV1CnnOracle(cfg)
    sd, osd = alg.networks.state_dict(), same_seed.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert all(torch.equal(sd[k], osd[k]) for k in sd)
    orc = DsacV1CnnOracle(cfg, state_dict=sd)
    for it in range(3):
        d = synth_image_batch(cfg, B, seed=it)
        torch.manual_seed(1000 + it)
        tb_ref = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise_v1(B, A), it)
        for k in V1_TB_KEYS[:-1]:
            assert float(tb_ref[k]) == float(tb[k]), k
        nets = alg.networks
        gref = torch.cat([p.grad.reshape(-1) for n in ("q", "policy") for p in getattr(nets, n).parameters()]
                         + [nets.log_alpha.grad.reshape(1)])
        assert torch.equal(gref, orc.flat_grads())
        sd, osd = nets.state_dict(), orc.state_dict()
        assert all(torch.equal(sd[k], osd[k]) for k in sd)
