"""value_hidden_sizes != policy_hidden_sizes (reference utils/common_utils.py:59-62 reads the two lists per key; a kwarg of SURVEY.md
section 8 rows a10 / a12): the policy nets get their own widths (`dsact_config.policy_hidden`) and, when the lists differ in length,
their own depth (`policy_n_hidden`), served by the tile-stage kernels -- the row-slice chains run one width per layer and one
layer count across all their units."""
import numpy as np
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hv,hp,B", [
    (24, 6, (64, 64), (32, 48), 64),                       # narrower policy
    (24, 6, (64, 64), (128, 96), 64),                      # wider policy
    (376, 17, (256, 256, 256), (128, 128, 128), 256),      # the BASELINE critics with a half-width policy
    (11, 3, (96, 40), (40, 96), 50),                       # ragged widths, odd batch
    (24, 6, (300, 64), (64, 300), 64),                     # more than one 256-chunk on one side only (row kernels' dispatch)
    # lists of different DEPTH (round 6, dsact_config.policy_n_hidden): a shallower / a deeper policy, one layer against four
    (24, 6, (64, 64, 64), (64, 64), 64),
    (24, 6, (64, 64), (48, 96, 32), 64),
    (376, 17, (256, 256, 256), (256, 256), 256),           # the BASELINE critics with a two-layer policy
    (11, 3, (40,), (96, 40, 24, 56), 50),
    (16, 4, (64, 48, 32, 64), (72,), 512),                 # split-K weight gradients
])
def test_unequal_hidden_sizes_against_the_oracle(O, A, hv, hp, B):
    """every intermediate, gradient, statistic and parameter against the oracle, which is pinned bit-exact to the live
    reference with these kwargs (tests/test_oracle_vs_reference.py::test_unequal_hidden_sizes_bit_exact_vs_live_reference)"""
    from test_hip_parity import run_case

    run_case("unequal widths O=%d A=%d value %s policy %s B=%d" % (O, A, hv, hp, B), O, A, hv, B, steps=3, policy_hidden_sizes=list(hp))


@pytest.mark.gpu
def test_unequal_widths_take_the_tile_stages_and_act_correctly():
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair

    O, A, hv, hp, B = 24, 6, (64, 64), (32, 48), 64
    alg, orc = make_pair(O, A, hv, B, seed=2, policy_hidden_sizes=list(hp))
    e = alg.engine
    assert not e.chain_active
    assert e.layout.n_pi == sum(o * i + o for o, i in ((32, O), (48, 32), (2 * A, 48)))
    obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
    want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
    np.testing.assert_allclose(e.policy_forward(obs), want, atol=2e-5, rtol=1e-5)                      # general path (3 rows)
    np.testing.assert_allclose(np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(3)]), want, atol=2e-5, rtol=1e-5)   # one-launch path
    # dsact_act_sample == TanhGaussDistribution.sample() on the same logits and generator state
    for i in range(5):
        torch.manual_seed(i)
        eps = torch.randn(1, A)
        action, logp = e.act_sample(obs[0], eps.numpy())
        dist = alg.networks.create_action_distributions(torch.from_numpy(e.policy_forward(obs[:1])))
        torch.manual_seed(i)
        a_ref, lp_ref = dist.sample()
        np.testing.assert_allclose(action, a_ref[0].numpy(), atol=2e-6, rtol=0)
        assert abs(float(logp[0]) - float(lp_ref[0])) <= 2e-4


@pytest.mark.gpu
def test_unequal_widths_graph_replays_equal_eager_updates():
    from test_hip_parity import make_pair

    O, A, hv, hp, B, N = 16, 4, (64, 64), (96, 32), 64, 2048
    algs = []
    for mode in ("eager", "graph"):
        alg, _ = make_pair(O, A, hv, B, seed=4, policy_hidden_sizes=list(hp))
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


@pytest.mark.gpu
def test_unequal_depths_act_and_replay_like_the_oracle(tmp_path):
    """lists of different length: the three acting forwards (tile stages, one launch, host) against the oracle's policy, graph
    replays == eager updates bitwise, and a checkpoint that loads into a reference-shaped CPU container"""
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair, hip_kwargs
    from dsac_v2_hip import ApproxContainer

    O, A, B, N = 16, 4, 64, 2048
    for hv, hp in (((64, 64, 64), (96, 32)), ((64,), (48, 64, 32))):
        alg, orc = make_pair(O, A, hv, B, seed=2, policy_hidden_sizes=list(hp))
        e = alg.engine
        assert not e.chain_active and e.debug_get("act_fast") == 1.0
        assert e.layout.n_pi == sum(o * i + o for o, i in zip(list(hp) + [2 * A], [O] + list(hp)))
        obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
        want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
        np.testing.assert_allclose(e.policy_forward(obs), want, atol=2e-5, rtol=1e-5)
        for mode in (1, 0):
            e.debug_set("host_act", mode)
            got = np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(3)])
            assert e.debug_get("act_host") == float(mode)
            np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
        algs = []
        for mode in ("eager", "graph"):
            a2, _ = make_pair(O, A, hv, B, seed=4, policy_hidden_sizes=list(hp))
            e2 = a2.engine
            e2.set_device_rng(777)
            e2.buffer_create(N)
            g = torch.Generator(device="cuda").manual_seed(1)
            e2.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                                  torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                                  (torch.rand(N, device="cuda", generator=g) < .05).float())
            np.random.seed(1)
            e2.upload_index_table(np.random.randint(0, N, size=(7, B)))
            if mode == "graph":
                e2.graph_build(4)
                e2.graph_run(1, 12)
            else:
                assert e2.time_steps(1, 12, use_graph=False) > 0
            e2.sync()
            algs.append(a2)
        for name in ("online", "target", "adam_m", "adam_v"):
            assert torch.equal(getattr(algs[0].engine, name), getattr(algs[1].engine, name)), name
        assert torch.isfinite(algs[1].engine.online).all()
        sd = algs[1].networks.state_dict()
        torch.save(sd, tmp_path / "apprfunc.pkl")
        cpu = ApproxContainer(**hip_kwargs(O, A, hv, B, policy_hidden_sizes=list(hp)))
        cpu.load_state_dict(torch.load(tmp_path / "apprfunc.pkl", map_location="cpu"))
        x = torch.randn(5, O)
        np.testing.assert_allclose(algs[1].networks.policy(x).numpy(), cpu.policy(x).detach().numpy(), atol=2e-5, rtol=1e-5)
