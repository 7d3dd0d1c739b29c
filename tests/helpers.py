"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from oracle.dsact_oracle import DsactOracle, default_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_CASES = ["tiny_l3", "tiny_l2", "pendulum", "fixed_alpha"]


def load_step_case(name):
    z = np.load(os.path.join(GOLDEN, "step_%s.npz" % name))
    O, A = int(z["cfg_obs_dim"]), int(z["cfg_act_dim"])
    cfg = default_config(
        O, A, hidden=[int(h) for h in z["cfg_hidden"]], act_limit=float(z["cfg_act_limit"]),
        auto_alpha=bool(int(z["cfg_auto_alpha"])), alpha=float(z["cfg_alpha"]),
        delay_update=int(z["cfg_delay_update"]))
    init = {k[len("init/"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("init/")}
    return z, cfg, init


def step_inputs(z, it):
    data = {k: torch.as_tensor(z["s%d/%s" % (it, k)]) for k in ("obs", "obs2", "act", "rew", "done", "logp")}
    noise = {k: torch.as_tensor(z["s%d/%s" % (it, k)]) for k in ("eps_new", "eps_2", "z5", "z6")}
    return data, noise


def synth_batch(rng, B, O, A, lim=0.4, p_done=0.01):
    return {
        "obs": torch.as_tensor(rng.standard_normal((B, O), dtype=np.float32)),
        "obs2": torch.as_tensor(rng.standard_normal((B, O), dtype=np.float32)),
        "act": torch.as_tensor(rng.uniform(-lim, lim, (B, A)).astype(np.float32)),
        "rew": torch.as_tensor(rng.standard_normal(B, dtype=np.float32)),
        "done": torch.as_tensor((rng.random(B) < p_done).astype(np.float32)),
        "logp": torch.zeros(B),
    }


def make_oracle(cfg, init=None, seed=0):
    torch.manual_seed(seed)
    return DsactOracle(cfg, state_dict=init)


def hip_kwargs(O, A, hidden, B, act_limit=0.4, **over):
    """The reference's flat kwargs dict (example_train/dsacv2_mlp_mujoco_offserial.py defaults +
    utils/init_args.py) for synthetic shapes, with the additive HIP keys."""
    kw = dict(
        algorithm="DSAC_V2_HIP", env_id="synthetic", seed=0, action_type="continu",
        value_func_name="ActionValueDistri", value_func_type="MLP",
        value_hidden_sizes=list(hidden), value_hidden_activation="gelu", value_output_activation="linear",
        policy_func_name="StochaPolicy", policy_func_type="MLP", policy_act_distribution="TanhGaussDistribution",
        policy_hidden_sizes=list(hidden), policy_hidden_activation="gelu", policy_output_activation="linear",
        policy_min_log_std=-20, policy_max_log_std=0.5,
        value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4,
        gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2,
        buffer_name="hip_replay_buffer", buffer_warm_size=1000, buffer_max_size=10000,
        replay_batch_size=B, obsv_dim=O, action_dim=A,
        action_high_limit=np.full((A,), act_limit, dtype=np.float32),
        action_low_limit=np.full((A,), -act_limit, dtype=np.float32),
        additional_info={}, cnn_shared=False, trainer="off_serial_trainer", use_gpu=True,
        # the parity cases pick ragged / unequal widths to reach the tile-stage kernels: they keep the exact arena layout;
        # tests/test_padded_widths.py turns the zero-padded storage (the plugin's default) on
        hip_pad_widths=False,
    )
    kw.update(over)
    return kw


def humanoid_digest():
    """tests/golden/step_humanoid_digest.npz (the BASELINE.json configuration run by the unmodified reference) plus
    the regenerated nets / minibatches / noise; checks the regeneration against the checksums the file holds."""
    from oracle.dsact_oracle import seeded_state_dict
    from oracle.make_golden import HUMANOID, humanoid_inputs

    z = np.load(os.path.join(GOLDEN, "step_humanoid_digest.npz"))
    H = HUMANOID
    assert [int(v) for v in z["cfg_seeds"]] == [H["init_seed"], H["batch_seed"], H["noise_seed0"]]
    cfg = default_config(H["O"], H["A"], hidden=list(H["hid"]), act_limit=H["lim"])
    template = DsactOracle(cfg).state_dict()
    init = seeded_state_dict(template, H["init_seed"])
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in init.values()], z["init_abs_sums"], rtol=1e-12)
    rng = np.random.default_rng(H["batch_seed"])
    steps = []
    for it in range(int(z["cfg_steps"])):
        b, noise = humanoid_inputs(it, rng)
        sums = [float(np.float64(b[k]).sum()) for k in ("obs", "obs2", "act", "rew", "done")] + \
               [float(noise[k].double().sum()) for k in ("eps_new", "eps_2", "z5", "z6")]
        np.testing.assert_allclose(sums, z["s%d/in_sums" % it], rtol=1e-12, atol=1e-12)
        steps.append(({k: torch.as_tensor(v) for k, v in b.items()}, noise))
    return z, cfg, init, steps


def policy_saturation_budget(orc, data, noise, B):
    """What fp32 itself does to the POLICY gradient when an action sits on its limit (the gradient-side twin of the
    log-prob budget in compare_intermediates): d logp / dx = 2 t (1 - t^2) / (1 + 1e-6 - t^2) with t = tanh(x) rounded to
    6e-8 and t^2 once more -- numerator and denominator each carry ~1.2e-7 of absolute noise (which of the two products
    is fused differs between ATen's kernels and any other evaluation order), i.e. a RELATIVE noise of 2.4e-7 / (1 + 1e-6 -
    t^2) on a factor of magnitude <= 2 that enters d loss / d mean as alpha / B and d loss / d std as alpha / B * |eps|.
    For every (row, dimension) where that exceeds 1e-5 the noise is pushed through the oracle's own policy net
    (|d logit / d theta| by autograd) -- an element-wise bound, zero for batches without saturated actions."""
    from oracle.dsact_oracle import policy_forward

    params = orc.p["policy"]
    logits = policy_forward(data["obs"], params, orc.cfg, None, (orc.act_sides or {}).get("pi"))
    A = logits.shape[1] // 2
    x = (logits[:, :A] + noise["eps_new"] * logits[:, A:]).detach().double()
    rel = 2.4e-7 / (1.0 + 1e-6 - torch.tanh(x) ** 2)
    d_mean = orc._alpha() / B * 2.0 * rel
    budget = [torch.zeros_like(p, dtype=torch.float64) for p in params]
    for b, k in (rel > 1e-5).nonzero().tolist():
        for col, w in ((k, float(d_mean[b, k])), (A + k, float(d_mean[b, k]) * abs(float(noise["eps_new"][b, k])))):
            for acc, g in zip(budget, torch.autograd.grad(logits[b, col], params, retain_graph=True, allow_unused=True)):
                if g is not None:   # (policy_std_type "parameter": the std column does not depend on the mean net)
                    acc += w * g.abs().double()
    if hasattr(orc, "arena_order"):   # arena order ("parameter": zero rows behind the mean rows of the output layer, log_std in the
        budget = orc.arena_order("policy", budget)   # bias tail; "mlp_separated": the twin-trunk layout)
    return torch.cat([t.reshape(-1).double() for t in budget]).numpy()
