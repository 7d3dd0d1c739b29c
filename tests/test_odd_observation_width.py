"""Observation widths that are no multiple of 4 on the row-slice chains (round 6). `obs_dim % 4 == 0` was a condition of the fast
path since round 2 -- HalfCheetah / Walker2d (17), Hopper (11), Ant (27 / 105), Pendulum (3) took the tile-stage kernels. The
only code behind it was the packed-copy store of a Q net's first layer (csrc/dsact_kernels.h mirror_store4 / mirror_fwd_each):
the quad straddling the observation / action boundary and the action columns behind it are placed element-wise now."""
import numpy as np
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B", [
    (3, 1, (64, 64, 64), 64),           # Pendulum
    (11, 3, (64, 64), 64),              # Hopper
    (17, 6, (256, 256, 256), 256),      # HalfCheetah / Walker2d at the BASELINE widths and batch
    (27, 8, (128, 128), 128),           # Ant (observation without contact forces)
    (105, 8, (256, 256), 512),          # Ant; 512 rows per weight-gradient tile
    (17, 6, (256, 256, 256), 1024),     # throughput-regime kernels
    (9, 2, (64,), 32),                  # one hidden layer, observation + action = 11 columns
    (5, 1, (128, 128, 128, 128), 16),   # four layers, the smallest batch the chains take
])
def test_odd_observation_widths_against_the_oracle(O, A, hid, B):
    from test_hip_parity import make_pair, run_case

    alg, _ = make_pair(O, A, hid, B)
    assert alg.engine.chain_active
    alg.engine.close()
    run_case("odd observation width O=%d A=%d %s B=%d" % (O, A, hid, B), O, A, hid, B, steps=3)


@pytest.mark.gpu
@pytest.mark.parametrize("O,A,hid,B", [(17, 6, (256, 256, 256), 256), (11, 3, (64, 64), 64), (3, 1, (64, 64), 64)])
def test_odd_observation_widths_graph_replays_equal_eager_updates(O, A, hid, B, monkeypatch):
    """the pipelined graph (fused optimiser tiles keep the packed copies fresh: the element-wise placement inside dw2_tile's
    mirror stores), the round-4 launch forms and the tile-stage kernels walk the same updates: graph == eager bitwise per form,
    chains == tile stages within parity tolerances"""
    from test_hip_parity import make_pair

    N = 4096
    out = {}
    for form in ("eager", "graph", "tiles"):
        if form == "tiles":
            monkeypatch.setenv("DSACT_NO_CHAIN", "1")
        alg, _ = make_pair(O, A, hid, B, seed=4)
        e = alg.engine
        assert e.chain_active == (form != "tiles")
        e.set_device_rng(777)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(7, B)))
        if form == "graph":
            e.graph_build(4)
            assert e.debug_get("pipe_graph") == 1.0
            e.graph_run(1, 12)
        else:
            assert e.time_steps(1, 12, use_graph=False) > 0
        e.sync()
        out[form] = {k: getattr(e, k).clone() for k in ("online", "target", "adam_m", "adam_v")}
        monkeypatch.delenv("DSACT_NO_CHAIN", raising=False)
    for k in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(out["eager"][k], out["graph"][k]), k
    np.testing.assert_allclose(out["graph"]["online"].cpu().numpy(), out["tiles"]["online"].cpu().numpy(), atol=3e-5, rtol=1e-4)
    assert torch.isfinite(out["graph"]["online"]).all()


@pytest.mark.gpu
def test_odd_observation_width_acting_and_groups():
    """acting forwards (tile stages / one launch / host) and a group of updates through the plugin surface at O = 17"""
    from oracle.dsact_oracle import policy_forward
    from test_hip_parity import make_pair
    from helpers import synth_batch

    O, A, hid, B = 17, 6, (256, 256, 256), 256
    alg, orc = make_pair(O, A, hid, B, seed=2)
    e = alg.engine
    obs = np.random.default_rng(0).standard_normal((3, O)).astype(np.float32)
    want = policy_forward(torch.as_tensor(obs), [p.detach() for p in orc.p["policy"]], orc.cfg).numpy()
    np.testing.assert_allclose(e.policy_forward(obs), want, atol=2e-5, rtol=1e-5)
    for mode in (1, 0):
        e.debug_set("host_act", mode)
        got = np.concatenate([e.policy_forward(obs[i:i + 1]) for i in range(3)])
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    rng = np.random.default_rng(3)
    for it in range(4):
        tb = alg.local_update(synth_batch(rng, B, O, A), it)
    assert np.isfinite(float(tb["Loss/Critic loss-RL iter"]))
