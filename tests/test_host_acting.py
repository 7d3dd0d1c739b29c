"""Host-side acting (csrc/dsact_host_act.h; SURVEY.md section 8 f1's "policy-weights snapshot for acting"; reference
training/off_sampler.py:46-56): dsact_act_sample / dsact_policy_forward(n = 1) on the calling thread from a pinned snapshot of
the policy net that is refreshed on the handle's stream behind every enqueued update that moves the policy.

  * same results as the one-launch GPU forward (csrc/dsact_act.h) within fp32 summation-order noise, for the shapes the
    acting path serves (BASELINE shape, ragged widths, unequal policy widths, policy_std_type = "parameter", GaussDistribution);
  * acts with the weights of the LAST COMPLETED update through every update entry point (eager step, compute_grads +
    apply_update, group replay, graph replay), and an update that leaves the policy alone (iteration % delay_update != 0)
    enqueues no copy;
  * load_state_dict and in-place torch writes to the policy's parameters reach the snapshot;
  * HipOffSampler's fast path runs on it (no launch per environment step) and equals the GPU-forward run from the same seeds.
"""
import numpy as np
import pytest
import torch

from helpers import hip_kwargs
from test_hip_parity import make_pair

pytestmark = pytest.mark.gpu


def both_forwards(e, obs, eps):
    """(action, logp, logits) through the host path and through the one-launch GPU forward on the same weights"""
    out = []
    for host in (1, 0):
        e.debug_set("host_act", host)
        assert e.debug_get("act_host") == float(host)
        a, lp = e.act_sample(obs, eps)
        out.append((a.copy(), float(lp[0]), e.policy_forward(obs[None])[0].copy()))
    e.debug_set("host_act", 1)
    return out


def close(h, g, lim, tag):
    scale = float(np.abs(g[2]).max()) + 1.0
    np.testing.assert_allclose(h[2], g[2], atol=4e-6 * scale, rtol=2e-5, err_msg=str(tag))
    np.testing.assert_allclose(h[0], g[0], atol=4e-6 * lim * scale, rtol=0, err_msg=str(tag))
    t2 = (np.asarray(g[0], np.float64) / lim) ** 2
    tol = 5e-4 + float((2.4e-7 / (1.0 + 1e-6 - np.minimum(t2, 1.0))).sum())
    assert abs(h[1] - g[1]) <= tol, (tag, h[1], g[1], tol)


@pytest.mark.parametrize("O,A,hid,B,lim,over", [
    (376, 17, (256, 256, 256), 256, 0.4, {}),
    (5, 1, (33,), 7, 2.0, {}),
    (24, 6, (128, 128), 64, 1.0, {"policy_std_type": "parameter"}),
    (24, 6, (64, 64), 64, 0.4, {"policy_act_distribution": "GaussDistribution"}),
    (16, 4, (64, 64), 64, 0.4, {"policy_hidden_sizes": [96, 40]}),
    (16, 4, (64, 64), 64, 0.4, {"policy_hidden_activation": "tanh"}),
    (24, 6, (64, 64), 64, 0.4, {"policy_output_activation": "tanh", "value_output_activation": "tanh"}),
    (24, 6, (64, 64), 64, 0.4, {"policy_output_activation": "sigmoid", "policy_std_type": "parameter"}),
])
def test_host_forward_equals_gpu_forward(O, A, hid, B, lim, over):
    alg, _ = make_pair(O, A, hid, B, act_limit=lim, seed=61, **over)
    e = alg.engine
    rng = np.random.default_rng(2)
    for i in range(12):
        obs = (3.0 * rng.standard_normal(O)).astype(np.float32)
        torch.manual_seed(i)
        eps = torch.randn(1, A).numpy()
        h, g = both_forwards(e, obs, eps)
        if over.get("policy_act_distribution") == "GaussDistribution":
            np.testing.assert_allclose(h[0], g[0], atol=1e-5, rtol=1e-5)
            assert abs(h[1] - g[1]) <= 5e-4
            np.testing.assert_allclose(h[2], g[2], atol=1e-5, rtol=2e-5)
        else:
            close(h, g, lim, (i, O, A, hid))
    assert e.debug_get("act_host_calls") >= 24


def test_snapshot_follows_every_update_entry_point():
    O, A, hid, B, N = 16, 4, (64, 64), 64, 1024
    alg, _ = make_pair(O, A, hid, B, seed=4)
    e = alg.engine
    e.set_device_rng(11)
    e.buffer_create(N)
    g = torch.Generator(device="cuda").manual_seed(1)
    e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                         torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                         (torch.rand(N, device="cuda", generator=g) < .05).float())
    np.random.seed(1)
    rows = np.random.randint(0, N, size=(8, B))
    e.upload_index_table(rows)
    obs = np.linspace(-1, 1, O).astype(np.float32)
    eps = np.full((1, A), 0.3, np.float32)

    def check(tag):
        h, gq = both_forwards(e, obs, eps)
        close(h, gq, 0.4, tag)
        return h[2]

    l0 = check("initial")
    e.act_sample(obs, eps)                       # (check() toggles the host path off and on, which marks the snapshot stale)
    copies = e.debug_get("act_copies")
    # an update that leaves the policy alone (iteration 1, delay_update 2): no copy is enqueued, the logits stay
    e.gather(rows[0]); e.step(1)
    e.act_sample(obs, eps)
    assert e.debug_get("act_copies") == copies
    assert np.array_equal(check("odd update"), l0)
    # one that moves it: exactly one copy, enqueued by the update call itself
    e.act_sample(obs, eps)
    copies = e.debug_get("act_copies")
    e.gather(rows[1]); e.step(2)
    assert e.debug_get("act_copies") == copies + 1
    e.act_sample(obs, eps)
    assert e.debug_get("act_copies") == copies + 1
    l1 = check("eager step")
    assert not np.array_equal(l1, l0)
    # gradient halves (the data-parallel seam)
    e.gather(rows[2]); e.compute_grads(4); e.apply_update(4)
    l2 = check("compute_grads + apply_update")
    assert not np.array_equal(l2, l1)
    # a group replay and a graph replay
    e.run_group(5, rows[:5])
    l3 = check("group replay")
    assert not np.array_equal(l3, l2)
    e.graph_build(4)
    e.graph_run(10, 8)
    l4 = check("graph replay")
    assert not np.array_equal(l4, l3)
    # torch writes: load_state_dict, and an in-place write under no_grad (the version counters the sampler looks at)
    sd = {k: v.clone() for k, v in alg.networks.state_dict().items()}
    for k in sd:
        if k.startswith("policy.policy.") and k.endswith("bias"):
            sd[k] += 0.05
    alg.networks.load_state_dict(sd)
    l5 = check("load_state_dict")
    assert not np.array_equal(l5, l4)
    with torch.no_grad():
        list(alg.networks.policy.parameters())[-1].add_(0.1)
    torch.cuda.synchronize()
    lg = alg.networks.policy(torch.from_numpy(obs)[None])[0].numpy()     # HipStochaPolicy.forward notices the version change
    assert np.abs(lg[:A] - l5[:A]).max() > 0.05
    check("in-place write")


def test_sampler_runs_on_the_host_forward():
    from test_hip_groups import _ToyEnv
    from training.hip_sampler import HipOffSampler

    outs = []
    for host in (True, False):
        alg, _ = make_pair(16, 4, (64, 64), 32, act_limit=0.3, seed=62, hip_host_act=host)
        e = alg.engine
        assert e.debug_get("act_host") == (1.0 if host else 0.0)
        smp = HipOffSampler(env=_ToyEnv(), networks=alg.networks, sample_batch_size=25, action_type="continu")
        torch.manual_seed(9)
        batch, _ = smp.sample()
        assert getattr(batch, "packed", None) is not None
        assert e.debug_get("act_host_calls") == (25.0 if host else 0.0)
        outs.append((batch, torch.randn(2)))
    (b0, r0), (b1, r1) = outs
    assert torch.equal(r0, r1)
    for s0, s1 in zip(b0, b1):
        np.testing.assert_allclose(s0[2], s1[2], atol=3e-6, rtol=0)          # actions
        np.testing.assert_allclose(s0[0], s1[0], atol=1e-5, rtol=0)          # the trajectories stay together over 25 steps
        assert abs(float(s0[6]) - float(s1[6])) <= 2e-3


def test_handoff_words_do_not_hide_each_other():
    """the hand-off timeout word was ONE word that update kernels (1) and the acting forward (2) overwrote -- an
    acting-forward timeout behind an update kernel's left `2`, and the failure was reported as benign. Two words now: the
    acting forward's alone fails the call only; the update kernels' (alone or with it) invalidates the state."""
    from dsact._ffi import DsactError

    alg, _ = make_pair(16, 4, (64, 64), 64, seed=4)
    e = alg.engine
    e.set_device_rng(5)
    st = e.get_state()
    e.debug_set("raise_handoff_word", 2)                     # the acting forward's word only
    with pytest.raises(DsactError, match="acting forward"):
        e.sync()
    assert e.debug_get("state_invalid") == 0.0
    e.sync()
    e.debug_set("raise_handoff_word", 3)                     # both: the update kernels' word decides
    with pytest.raises(DsactError, match="restore them"):
        e.sync()
    assert e.debug_get("state_invalid") == 1.0
    e.set_state(adam_steps=st["adam_steps"], mean_std=st["mean_std"])
    assert e.debug_get("state_invalid") == 0.0
    e.debug_set("raise_handoff_word", 1)
    with pytest.raises(DsactError, match="restore them"):
        e.sync()
    e.debug_set("ack_state", 1)
    e.sync()
