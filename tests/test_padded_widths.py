"""Ragged / unequal hidden widths on the row-slice chain kernels (round 6): the chains run ONE width per layer
across all their units, so such nets are STORED zero-padded to 64 / 128 / 256 (dsac-v2_amd/dsact/layout.py ArenaLayout pad_to,
dsact/engine.py pad_widths; the plugin's default `hip_pad_widths=True`). The reference's tensors are the top-left windows of the
stored ones; the padding is structurally zero and every gradient element that touches it is an exact 0, so it stays zero."""
import numpy as np
import pytest
import torch

from dsact.engine import DsactEngine
from dsact.layout import ArenaLayout


def test_padded_layout_windows_and_the_padding_rule():
    O, A = 24, 6
    for kw in (dict(), dict(policy_std_type="parameter")):
        lay = ArenaLayout(O, A, [96, 40], policy_hidden=[40, 100], pad_to=128, **kw)
        full = ArenaLayout(O, A, [128, 128], **kw)
        assert (lay.n_q, lay.n_pi, lay.n_online, lay.n_target) == (full.n_q, full.n_pi, full.n_online, full.n_target)
        ref = ArenaLayout(O, A, [96, 40], policy_hidden=[40, 100], **kw)
        assert list(lay.state_dict_keys().items()) == list(ref.state_dict_keys().items())        # the reference's names and shapes
        assert lay.flop_per_step(64) == ref.flop_per_step(64)                                      # algorithmic cost: the logical nets
        big = {(n, v[0]): v for n in full.all_nets for v in full.param_views(n)}
        flat = torch.arange(lay.n_online)
        seen = np.zeros(lay.n_online, np.int32)
        for n in lay.all_nets:
            for name, arena, off, shape, strides in lay.param_views(n):
                w = big[(n, name)]
                assert (arena, off, strides) == (w[1], w[2], w[4]) and all(a <= b for a, b in zip(shape, w[3])), (n, name)
                if arena == "online":
                    seen[torch.as_strided(flat, shape, strides, off).reshape(-1).numpy()] += 1
        assert seen.max() == 1
        assert lay.zero_rows("policy") == full.zero_rows("policy")
    P = DsactEngine._pad_width
    assert P([96, 40], None, 64, 24) == 128 and P([64, 64], [32, 48], 64, 24) == 64 and P([256] * 3, [128] * 3, 256, 376) == 256
    assert P([200] * 3, None, 1024, 376) == 256
    assert P([256] * 3, None, 256, 376) is None and P([64, 64], None, 64, 24) is None            # the chains take these as they are
    assert P([300, 64], None, 64, 24) is None                                                   # wider than the chains go
    assert P([96, 40], None, 50, 24) is None and P([96, 40], None, 64, 11) == 128               # a batch the chains refuse; any observation width
    assert P([96, 40], None, 64, 24, value_act=4) is None and P([96, 40], None, 64, 24, policy_act=4) is None   # sigmoid(0) != 0
    assert P([96, 40], [40], 64, 24) is None and P([96, 40], None, 64, 24, policy_std_type="mlp_separated") is None
    assert P([96, 40], None, 64, 24, algo="DSAC_V1") == 128                                      # DSAC_V1 too


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B,over", [
    (24, 6, (96, 40), 64, {}),                                                  # ragged -> 128
    (24, 6, (64, 64), 64, {"policy_hidden_sizes": [32, 48]}),                   # narrower policy -> 64
    (24, 6, (64, 64), 64, {"policy_hidden_sizes": [128, 96]}),                  # wider policy -> 128 (the critics are padded)
    (376, 17, (256, 256, 256), 256, {"policy_hidden_sizes": [128, 128, 128]}),  # the BASELINE critics with a half-width policy
    (376, 17, (200, 200, 200), 256, {}),                                        # the BASELINE shape at 200 wide -> 256
    (16, 4, (100, 100), 64, {"value_hidden_activation": "relu", "policy_hidden_activation": "tanh"}),
    (16, 4, (100, 72), 64, {"value_hidden_activation": "elu", "policy_hidden_activation": "selu"}),
    (24, 6, (96, 40), 64, {"policy_std_type": "parameter"}),
    (24, 6, (96, 40), 64, {"value_output_activation": "tanh", "policy_output_activation": "tanh"}),
    (24, 6, (96, 40), 64, {"policy_act_distribution": "GaussDistribution"}),
    (376, 17, (200, 200, 200), 1024, {}),                                       # throughput-regime kernels
    (24, 6, (96, 40), 512, {}),                                                 # 512 rows per weight-gradient tile
])
def test_padded_widths_against_the_oracle(O, A, hid, B, over):
    """every intermediate (cut to the reference's widths; the padded features must read exact zeros), gradient, statistic and
    parameter of three updates against the oracle -- whose flat views are padded the same way -- on the row-slice chains"""
    from test_hip_parity import make_pair, run_case

    alg, _ = make_pair(O, A, hid, B, hip_pad_widths=True, **over)
    e = alg.engine
    assert e.chain_active and e.layout.pad_to in (64, 128, 256)
    alg.engine.close()
    run_case("padded widths O=%d A=%d hid=%s B=%d %s" % (O, A, hid, B, over), O, A, hid, B, steps=3, hip_pad_widths=True, **over)


@pytest.mark.gpu
def test_sigmoid_nets_and_refused_shapes_keep_the_exact_layout():
    from test_hip_parity import make_pair

    alg, _ = make_pair(24, 6, (96, 40), 64, hip_pad_widths=True, value_hidden_activation="sigmoid")
    assert alg.engine.layout.pad_to is None and not alg.engine.chain_active
    alg, _ = make_pair(11, 3, (96, 40), 50, hip_pad_widths=True)     # batch 50: no multiple of 16
    assert alg.engine.layout.pad_to is None and not alg.engine.chain_active
    alg, _ = make_pair(24, 6, (64, 64), 64, hip_pad_widths=True)
    assert alg.engine.layout.pad_to is None and alg.engine.chain_active


@pytest.mark.gpu
def test_padding_stays_zero_through_graph_replays_checkpoints_and_acting(tmp_path):
    """pipelined graph replays == eager updates bitwise; everything outside the reference's windows is an exact zero in online,
    target and both Adam moments afterwards; state_dict has the reference's shapes and loads into a CPU container whose forward
    agrees with the three acting forwards; the trajectory equals the exact-layout (tile-stage) one within parity tolerances"""
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair, hip_kwargs
    from dsac_v2_hip import ApproxContainer

    O, A, hv, hp, B, N = 16, 4, (96, 40), (40, 100), 64, 2048
    algs = {}
    for mode, pad in (("eager", True), ("graph", True), ("exact", False)):
        alg, orc = make_pair(O, A, hv, B, seed=4, policy_hidden_sizes=list(hp), hip_pad_widths=pad)
        e = alg.engine
        assert (e.layout.pad_to == 128 and e.chain_active) if pad else (e.layout.pad_to is None and not e.chain_active)
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
        algs[mode] = alg
    e0, e1 = algs["eager"].engine, algs["graph"].engine
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(e0, name), getattr(e1, name)), name
    lay = e1.layout
    for arena_name, tensors in (("online", (e1.online, e1.adam_m, e1.adam_v)), ("target", (e1.target,))):
        live = torch.zeros(tensors[0].numel(), dtype=torch.bool, device="cuda")
        for net in lay.all_nets:
            for _, arena, off, shape, strides in lay.param_views(net):
                if arena == arena_name:
                    torch.as_strided(live, shape, strides, off).fill_(True)
        if arena_name == "online":
            live[lay.log_alpha_offset] = True
        for t in tensors:
            assert not t[:live.numel()][~live].any(), arena_name
    assert torch.isfinite(e1.online).all()
    # same device noise, same rows: the padded chains and the exact-layout tile stages walk the same trajectory
    sd, sx = algs["graph"].networks.state_dict(), algs["exact"].networks.state_dict()
    assert list(sd.keys()) == list(sx.keys())
    for k in sd:
        assert sd[k].shape == sx[k].shape
        np.testing.assert_allclose(sd[k].cpu().numpy(), sx[k].cpu().numpy(), atol=3e-5, rtol=1e-4, err_msg=k)
    assert sd["policy.policy.2.weight"].shape == (100, 40) and sd["q1.q.2.weight"].shape == (40, 96)
    torch.save(sd, tmp_path / "apprfunc.pkl")
    cpu = ApproxContainer(**hip_kwargs(O, A, hv, B, policy_hidden_sizes=list(hp)))
    cpu.load_state_dict(torch.load(tmp_path / "apprfunc.pkl", map_location="cpu"))
    obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
    want = cpu.policy(torch.from_numpy(obs)).detach().numpy()
    np.testing.assert_allclose(e1.policy_forward(obs), want, atol=2e-5, rtol=1e-5)
    for mode in (1, 0):
        e1.debug_set("host_act", mode)
        got = np.concatenate([e1.policy_forward(obs[i:i + 1]) for i in range(3)])
        assert e1.debug_get("act_host") == float(mode)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    # the optimiser sidecar holds the moment ARENAS: one written by the other layout is refused by name, its own round-trips
    with pytest.raises(ValueError, match="hip_pad_widths"):
        algs["graph"].load_optimizer_state_dict(algs["exact"].optimizer_state_dict())
    algs["graph"].load_optimizer_state_dict(algs["graph"].optimizer_state_dict())
    # a checkpoint written by the exact layout loads into the padded one (and back): windows only, the padding is untouched
    algs["graph"].networks.load_state_dict({k: v.cpu() for k, v in sx.items()})
    e1.sync()
    for k, v in algs["graph"].networks.state_dict().items():
        assert torch.equal(v.cpu(), sx[k].cpu()), k
