"""The oracle restatement vs the golden vectors generated from the unmodified reference
(oracle/make_golden.py). Runs anywhere (no GPU, no /root/reference needed)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.dsact_oracle import TB_KEYS, DsactOracle, MT19937, ReplayOracle, randint_legacy
from helpers import GOLDEN, STEP_CASES, humanoid_digest, load_step_case, step_inputs


@pytest.mark.parametrize("name", STEP_CASES)
def test_step_matches_reference_golden(name):
    torch.set_num_threads(1)
    z, cfg, init = load_step_case(name)
    orc = DsactOracle(cfg, state_dict=init)
    for it in range(int(z["cfg_steps"])):
        data, noise = step_inputs(z, it)
        tb = orc.local_update(data, noise, it)
        got = np.array([float(tb[k]) for k in TB_KEYS[:-1]])
        # same torch build => bit-exact; a different torch build may differ in the last ulp
        np.testing.assert_allclose(got, z["s%d/tb" % it], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(orc.flat_grads().numpy(), z["s%d/grad" % it], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(orc.flat_params().numpy(), z["s%d/params" % it], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc.flat_targets().numpy(), z["s%d/targets" % it], rtol=0, atol=2e-6)


def test_humanoid_b256_matches_reference_digest():
    """the BASELINE.json configuration (obs 376 / act 17, 3x256, batch 256): oracle vs the digests the unmodified
    reference produced from the same seeded nets, minibatches and noise"""
    torch.set_num_threads(1)
    z, cfg, init, steps = humanoid_digest()
    orc = DsactOracle(cfg, state_dict=init)
    stride = int(z["cfg_stride"])
    for it, (data, noise) in enumerate(steps):
        tb = orc.local_update(data, noise, it)
        got = np.array([float(tb[k]) for k in TB_KEYS[:-1]])
        np.testing.assert_allclose(got, z["s%d/tb" % it], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(orc.flat_grads().numpy()[::stride], z["s%d/grad_s" % it], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(orc.flat_params().numpy()[::stride], z["s%d/params_s" % it], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc.flat_targets().numpy()[::stride], z["s%d/targets_s" % it], rtol=0, atol=2e-6)
        # the arena's last element (log_alpha) is not on the stride: pinned explicitly
        np.testing.assert_allclose(orc.flat_grads().numpy()[-1:], z["s%d/grad_log_alpha" % it], rtol=1e-6)
        np.testing.assert_allclose(orc.flat_params().numpy()[-1:], z["s%d/log_alpha" % it], rtol=0, atol=1e-7)
        l2 = [float(g.double().norm()) for n in ("q1", "q2", "policy") for g in (p.grad for p in orc.p[n])]
        np.testing.assert_allclose(l2, z["s%d/grad_l2" % it], rtol=1e-5)


def test_state_dict_layout_matches_reference():
    lay = json.load(open(os.path.join(GOLDEN, "checkpoint_layout.json")))
    from oracle.dsact_oracle import default_config
    orc = DsactOracle(default_config(376, 17, (256, 256, 256)))
    got = [[k, list(v.shape)] for k, v in orc.state_dict().items()]
    assert got == lay["humanoid_l3"]
    orc = DsactOracle(default_config(3, 1, (256, 256, 256), act_limit=2.0))
    got = [[k, list(v.shape)] for k, v in orc.state_dict().items()]
    assert got == lay["pendulum_shipped"]
    # the shipped apprfunc_0.pkl was saved after the iteration-0 update: log_alpha = 1 - lr_alpha
    assert abs(lay["pendulum_shipped_log_alpha_it0"] - (1.0 - 3e-4)) < 1e-6


def test_replay_ring_and_gather_match_reference_golden():
    z = np.load(os.path.join(GOLDEN, "replay.npz"))
    O, A, N = [int(v) for v in z["cfg"]]
    buf = ReplayOracle(O, A, N)
    samples = []
    for i in range(73):
        r, d, l = z["in/rdl%d" % i]
        samples.append((z["in/obs%d" % i], {}, z["in/act%d" % i], float(r), z["in/obs2_%d" % i], bool(d),
                        np.float32(l), {}))
    buf.add_batch(samples[:30])
    assert (buf.size, buf.ptr) == (int(z["size_30"]), int(z["ptr_30"]))
    np.random.seed(11)
    b = buf.sample_batch(16)
    for k, v in b.items():
        np.testing.assert_array_equal(v.numpy(), z["b30/" + k])
    buf.add_batch(samples[30:])
    assert (buf.size, buf.ptr) == (int(z["size_73"]), int(z["ptr_73"]))
    b = buf.sample_batch(16)
    for k, v in b.items():
        np.testing.assert_array_equal(v.numpy(), z["b73/" + k])


@pytest.mark.parametrize("n", [1, 2, 10000, 12345, 2 ** 20, 2 ** 20 + 1, 10 ** 6, 10 ** 7])
def test_index_draw_restatement_bit_exact(n):
    z = np.load(os.path.join(GOLDEN, "replay.npz"))
    want = z["idx/%d" % n]
    got = randint_legacy(MT19937(1), n, 700)
    np.testing.assert_array_equal(got, want)
    np.random.seed(1)
    np.testing.assert_array_equal(np.random.randint(0, n, size=700), want)


def test_cnn_step_matches_reference_golden():
    """CNN approximators (networks/cnn.py, conv_type type_2): digests recorded from the unmodified
    reference by oracle/make_golden.py::gen_cnn_case; nets and minibatches regenerate from seeds."""
    from oracle.dsact_oracle import draw_noise
    from oracle.dsact_oracle_cnn import DsactCnnOracle, cnn_config, synth_image_batch

    torch.set_num_threads(2)
    z = np.load(os.path.join(GOLDEN, "step_cnn_type2.npz"))
    cfg = cnn_config(tuple(int(v) for v in z["cfg_obs_shape"]), int(z["cfg_act_dim"]), str(z["cfg_conv_type"]))
    torch.manual_seed(0)
    orc = DsactCnnOracle(cfg)
    sd = orc.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    assert [str(list(v.shape)) for v in sd.values()] == [str(s) for s in z["shapes"]]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], z["init_sums"], rtol=1e-12, atol=1e-12)
    B, A = int(z["cfg_batch"]), cfg["act_dim"]
    for it in range(int(z["cfg_steps"])):
        d = synth_image_batch(cfg, B, seed=it)
        torch.manual_seed(1000 + it)
        tb = orc.local_update(d, draw_noise(B, A), it)
        got = np.array([float(tb[k]) for k in TB_KEYS[:-1]])
        np.testing.assert_allclose(got, z["s%d/tb" % it], rtol=2e-6, atol=2e-6)
        online = orc.p["q1"] + orc.p["q2"] + orc.p["policy"]
        np.testing.assert_allclose([float(p.grad.double().sum()) for p in online], z["s%d/grad_sum" % it], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose([float(p.grad.double().norm()) for p in online], z["s%d/grad_l2" % it], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose([float(v.double().sum()) for v in orc.state_dict().values()], z["s%d/param_sums" % it],
                                   rtol=1e-7, atol=1e-5)


def test_v1_step_matches_reference_golden():
    """DSAC_V1 restatement (oracle/dsac_v1_oracle.py) vs vectors recorded from the unmodified reference."""
    from oracle.dsact_oracle import default_config
    from oracle.dsac_v1_oracle import V1_TB_KEYS, DsacV1Oracle

    torch.set_num_threads(1)
    z = np.load(os.path.join(GOLDEN, "step_v1_tiny.npz"))
    cfg = default_config(int(z["cfg_obs_dim"]), int(z["cfg_act_dim"]), [int(h) for h in z["cfg_hidden"]],
                         act_limit=float(z["cfg_act_limit"]), TD_bound=float(z["cfg_td_bound"]))
    init = {k[len("init/"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("init/")}
    orc = DsacV1Oracle(cfg, state_dict=init)
    for it in range(int(z["cfg_steps"])):
        data = {k: torch.as_tensor(z["s%d/%s" % (it, k)]) for k in ("obs", "obs2", "act", "rew", "done")}
        noise = {k: torch.as_tensor(z["s%d/%s" % (it, k)]) for k in ("eps_new", "eps_2", "z_t")}
        tb = orc.local_update(data, noise, it)
        np.testing.assert_allclose([float(tb[k]) for k in V1_TB_KEYS[:-1]], z["s%d/tb" % it], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(orc.flat_grads().numpy(), z["s%d/grad" % it], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(orc.flat_params().numpy(), z["s%d/params" % it], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc.flat_targets().numpy(), z["s%d/targets" % it], rtol=0, atol=2e-6)
