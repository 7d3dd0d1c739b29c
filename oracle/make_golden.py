"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_loader.py) in this container.

    python oracle/make_golden.py

The fixtures are small (tiny nets) so they can be committed; they pin oracle/dsact_oracle.py to the
reference on boxes where /root/reference is absent (the GPU box). Each file records the torch / numpy
versions the reference ran under.

What is stored per case (`step_<name>.npz`):
  cfg_*                 shapes and hyper-parameters
  init/<key>            ApproxContainer.state_dict() before the first update (SURVEY.md App. C order)
  s<k>/obs,act,...      the minibatch fed to DSAC_V2.local_update at iteration k
  s<k>/eps_new,...      the 8 torch.randn draws the reference consumes in that call (App. A.1 order)
  s<k>/tb               the 14 numeric tb_info entries (dsac_v2.py:188-204)
  s<k>/grad             flat gradient [q1 | q2 | policy | log_alpha] after __compute_gradient
  s<k>/params,targets   flat online / target parameters after __update
`replay.npz`: reference ReplayBuffer ring semantics + np.random.randint index draws.
`step_humanoid_digest.npz`: the BASELINE.json configuration itself (obs 376 / act 17, 3x256, batch 256), 4 updates of
the unmodified reference. 1.2 M parameters per step would not be a small fixture, so nets, minibatches and noise
regenerate from seeds (oracle.dsact_oracle.seeded_state_dict / synth_batch / draw_noise) and the file keeps digests:
tb_info, every DIGEST_STRIDE-th element of the flat gradient / parameter / target arenas, per-net max |g|, per-tensor
gradient L2 norms and parameter abs-sums, and checksums of the regenerated inputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle.dsact_oracle import TB_KEYS, draw_noise, seeded_state_dict  # noqa: E402

CASES = {
    # name: (obs, act, hidden, batch, act_limit, steps, extra kwargs)
    "tiny_l3": (11, 3, (32, 32, 32), 32, 0.4, 5, {}),
    "tiny_l2": (5, 2, (64, 64), 64, 1.0, 4, {}),
    "pendulum": (3, 1, (64, 64, 64), 64, 2.0, 4, {}),
    "fixed_alpha": (7, 2, (32, 32), 32, 0.4, 3, {"auto_alpha": False, "alpha": 0.2, "delay_update": 1}),
}


def synth_batch(rng, B, O, A, lim):
    return {
        "obs": rng.standard_normal((B, O), dtype=np.float32),
        "obs2": rng.standard_normal((B, O), dtype=np.float32),
        "act": rng.uniform(-lim, lim, (B, A)).astype(np.float32),
        "rew": rng.standard_normal(B, dtype=np.float32),
        "done": (rng.random(B) < 0.05).astype(np.float32),
        "logp": np.zeros(B, np.float32),
    }


def flat(params):
    return torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy()


def gen_step_case(ref, name, spec, out_dir):
    O, A, hid, B, lim, steps, extra = spec
    kw = ref_loader.reference_kwargs(O, A, hid, act_limit=lim, **extra)
    torch.manual_seed(1234)
    alg = ref.DSAC_V2(**kw)
    nets = alg.networks
    out = {
        "cfg_obs_dim": O, "cfg_act_dim": A, "cfg_hidden": np.array(hid), "cfg_batch": B,
        "cfg_act_limit": lim, "cfg_steps": steps,
        "cfg_auto_alpha": int(kw["auto_alpha"]), "cfg_alpha": kw["alpha"],
        "cfg_delay_update": kw["delay_update"],
        "versions": np.array([torch.__version__, np.__version__]),
    }
    for k, v in nets.state_dict().items():
        out["init/" + k] = v.numpy().copy()
    rng = np.random.default_rng(7)
    for it in range(steps):
        b = synth_batch(rng, B, O, A, lim)
        torch.manual_seed(1000 + it)
        noise = draw_noise(B, A)  # exactly what the reference will draw after the same seed
        torch.manual_seed(1000 + it)
        tb = alg.local_update({k: torch.as_tensor(v) for k, v in b.items()}, it)
        for k, v in b.items():
            out["s%d/%s" % (it, k)] = v
        for k in ("eps_new", "eps_2", "z5", "z6"):
            out["s%d/%s" % (it, k)] = noise[k].numpy()
        out["s%d/tb" % it] = np.array([float(tb[k]) for k in TB_KEYS[:-1]], np.float64)
        online = list(nets.q1.parameters()) + list(nets.q2.parameters()) + list(nets.policy.parameters())
        g = [p.grad for p in online]
        ga = nets.log_alpha.grad if nets.log_alpha.grad is not None else torch.zeros(())
        out["s%d/grad" % it] = torch.cat([x.reshape(-1) for x in g] + [ga.reshape(1)]).numpy().copy()
        out["s%d/params" % it] = np.concatenate([flat(online), nets.log_alpha.detach().reshape(1).numpy()])
        targets = (list(nets.q1_target.parameters()) + list(nets.q2_target.parameters())
                   + list(nets.policy_target.parameters()))
        out["s%d/targets" % it] = flat(targets)
    np.savez_compressed(os.path.join(out_dir, "step_%s.npz" % name), **out)
    print("wrote step_%s.npz" % name)


DIGEST_STRIDE = 499
HUMANOID = dict(O=376, A=17, hid=(256, 256, 256), B=256, lim=0.4, steps=4, init_seed=20240, batch_seed=7, noise_seed0=1000)


def humanoid_inputs(it, rng):
    """(minibatch, noise) of update `it` of the digest case; rng: np.random.default_rng(HUMANOID['batch_seed']),
    advanced by the caller in iteration order"""
    H = HUMANOID
    b = synth_batch(rng, H["B"], H["O"], H["A"], H["lim"])
    torch.manual_seed(H["noise_seed0"] + it)
    return b, draw_noise(H["B"], H["A"])


def gen_humanoid_digest(ref, out_dir):
    H = HUMANOID
    kw = ref_loader.reference_kwargs(H["O"], H["A"], H["hid"], act_limit=H["lim"])
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    nets = alg.networks
    nets.load_state_dict(seeded_state_dict(nets.state_dict(), H["init_seed"]))
    out = {"cfg_obs_dim": H["O"], "cfg_act_dim": H["A"], "cfg_hidden": np.array(H["hid"]), "cfg_batch": H["B"],
           "cfg_act_limit": H["lim"], "cfg_steps": H["steps"], "cfg_stride": DIGEST_STRIDE,
           "cfg_seeds": np.array([H["init_seed"], H["batch_seed"], H["noise_seed0"]]),
           "versions": np.array([torch.__version__, np.__version__]),
           "init_abs_sums": np.array([float(v.double().abs().sum()) for v in nets.state_dict().values()])}
    rng = np.random.default_rng(H["batch_seed"])
    for it in range(H["steps"]):
        b, noise = humanoid_inputs(it, rng)
        out["s%d/in_sums" % it] = np.array([float(np.float64(b[k]).sum()) for k in ("obs", "obs2", "act", "rew", "done")]
                                           + [float(noise[k].double().sum()) for k in ("eps_new", "eps_2", "z5", "z6")])
        torch.manual_seed(H["noise_seed0"] + it)   # the reference draws the same stream draw_noise just did
        tb = alg.local_update({k: torch.as_tensor(v) for k, v in b.items()}, it)
        out["s%d/tb" % it] = np.array([float(tb[k]) for k in TB_KEYS[:-1]], np.float64)
        groups = (list(nets.q1.parameters()), list(nets.q2.parameters()), list(nets.policy.parameters()))
        online = [p for g in groups for p in g]
        ga = nets.log_alpha.grad if nets.log_alpha.grad is not None else torch.zeros(())
        grad = torch.cat([p.grad.reshape(-1) for p in online] + [ga.reshape(1)]).numpy()
        params = np.concatenate([flat(online), nets.log_alpha.detach().reshape(1).numpy()])
        targets = flat(list(nets.q1_target.parameters()) + list(nets.q2_target.parameters())
                       + list(nets.policy_target.parameters()))
        out["s%d/grad_s" % it] = grad[::DIGEST_STRIDE].copy()
        out["s%d/params_s" % it] = params[::DIGEST_STRIDE].copy()
        out["s%d/targets_s" % it] = targets[::DIGEST_STRIDE].copy()
        # the stride misses the arena's last element: log_alpha's gradient and value are stored on their own
        out["s%d/grad_log_alpha" % it] = grad[-1:].copy()
        out["s%d/log_alpha" % it] = params[-1:].copy()
        out["s%d/grad_max" % it] = np.array([max(float(p.grad.abs().max()) for p in g) for g in groups] + [abs(float(ga))])
        out["s%d/grad_l2" % it] = np.array([float(p.grad.double().norm()) for p in online])
        out["s%d/param_abs_sums" % it] = np.array([float(p.detach().double().abs().sum()) for p in online])
    np.savez_compressed(os.path.join(out_dir, "step_humanoid_digest.npz"), **out)
    print("wrote step_humanoid_digest.npz")


def gen_replay(out_dir):
    import importlib

    rb_mod = importlib.import_module("training.replay_buffer")
    O, A, N = 4, 2, 50
    kw = dict(trainer="off_serial_trainer", seed=0, obsv_dim=O, action_dim=A, buffer_max_size=N,
              additional_info={})
    buf = rb_mod.ReplayBuffer(**kw)
    rng = np.random.default_rng(3)
    out = {"cfg": np.array([O, A, N])}
    samples = []
    for i in range(73):  # wraps the ring once
        s = (rng.standard_normal(O).astype(np.float32), {}, rng.uniform(-1, 1, A).astype(np.float32),
             float(rng.standard_normal()), rng.standard_normal(O).astype(np.float32), bool(rng.random() < 0.2),
             np.float32(rng.standard_normal()), {})
        samples.append(s)
    buf.add_batch(samples[:30])
    out["size_30"], out["ptr_30"] = buf.size, buf.ptr
    np.random.seed(11)
    b = buf.sample_batch(16)
    for k, v in b.items():
        out["b30/" + k] = v.numpy()
    buf.add_batch(samples[30:])
    out["size_73"], out["ptr_73"] = buf.size, buf.ptr
    b = buf.sample_batch(16)
    for k, v in b.items():
        out["b73/" + k] = v.numpy()
    for i, s in enumerate(samples):
        out["in/obs%d" % i], out["in/act%d" % i], out["in/obs2_%d" % i] = s[0], s[2], s[4]
        out["in/rdl%d" % i] = np.array([s[3], float(s[5]), float(s[6])], np.float64)
    # index draws: np.random.seed(seed); np.random.randint(0, n, size)
    for n in (1, 2, 10000, 12345, 2 ** 20, 2 ** 20 + 1, 10 ** 6, 10 ** 7):
        np.random.seed(1)
        out["idx/%d" % n] = np.random.randint(0, n, size=700)  # crosses an MT19937 twist boundary
    out["versions"] = np.array([torch.__version__, np.__version__])
    np.savez_compressed(os.path.join(out_dir, "replay.npz"), **out)
    print("wrote replay.npz")


def gen_checkpoint_layout(ref, out_dir):
    """Key/shape list of the shipped Pendulum checkpoints (results/.../apprfunc/*.pkl) and of a
    Humanoid-shaped container -- pins the state_dict wire format."""
    import json

    lay = {}
    p = os.path.join(ref_loader.REFERENCE_ROOT, "results/DSAC_V2_gym_pendulum/240223-003213/apprfunc")
    sd = torch.load(os.path.join(p, "apprfunc_0.pkl"), weights_only=True)
    lay["pendulum_shipped"] = [[k, list(v.shape)] for k, v in sd.items()]
    lay["pendulum_shipped_log_alpha_it0"] = float(sd["log_alpha"])
    kw = ref_loader.reference_kwargs(376, 17, (256, 256, 256))
    nets = ref.ApproxContainer(**kw)
    lay["humanoid_l3"] = [[k, list(v.shape)] for k, v in nets.state_dict().items()]
    with open(os.path.join(out_dir, "checkpoint_layout.json"), "w") as f:
        json.dump(lay, f, indent=1)
    print("wrote checkpoint_layout.json")


def gen_v1_case(out_dir):
    """DSAC_V1 (reference dsac_v1.py), tiny nets, same file format as step_<name>.npz"""
    import importlib

    from oracle.dsac_v1_oracle import V1_TB_KEYS, draw_noise_v1

    v1 = importlib.import_module("dsac_v1")
    O, A, hid, B, lim, steps = 9, 2, (32, 32), 32, 0.7, 4
    kw = ref_loader.reference_kwargs(O, A, hid, act_limit=lim, algorithm="DSAC_V1", TD_bound=10)
    torch.manual_seed(4321)
    alg = v1.DSAC_V1(**kw)
    nets = alg.networks
    out = {"cfg_obs_dim": O, "cfg_act_dim": A, "cfg_hidden": np.array(hid), "cfg_batch": B, "cfg_act_limit": lim,
           "cfg_steps": steps, "cfg_td_bound": 10.0, "versions": np.array([torch.__version__, np.__version__])}
    for k, v in nets.state_dict().items():
        out["init/" + k] = v.numpy().copy()
    rng = np.random.default_rng(17)
    for it in range(steps):
        b = synth_batch(rng, B, O, A, lim)
        torch.manual_seed(2000 + it)
        noise = draw_noise_v1(B, A)
        torch.manual_seed(2000 + it)
        tb = alg.local_update({k: torch.as_tensor(v) for k, v in b.items()}, it)
        for k, v in b.items():
            out["s%d/%s" % (it, k)] = v
        for k in ("eps_new", "eps_2", "z_t"):
            out["s%d/%s" % (it, k)] = noise[k].numpy()
        out["s%d/tb" % it] = np.array([float(tb[k]) for k in V1_TB_KEYS[:-1]], np.float64)
        online = list(nets.q.parameters()) + list(nets.policy.parameters())
        ga = nets.log_alpha.grad if nets.log_alpha.grad is not None else torch.zeros(())
        out["s%d/grad" % it] = torch.cat([p.grad.reshape(-1) for p in online] + [ga.reshape(1)]).numpy().copy()
        out["s%d/params" % it] = np.concatenate([flat(online), nets.log_alpha.detach().reshape(1).numpy()])
        out["s%d/targets" % it] = flat(list(nets.q_target.parameters()) + list(nets.policy_target.parameters()))
    np.savez_compressed(os.path.join(out_dir, "step_v1_tiny.npz"), **out)
    print("wrote step_v1_tiny.npz")


def gen_cnn_case(ref, out_dir):
    """CNN approximators: nets and minibatches regenerate from seeds
    (torch.manual_seed / numpy default_rng are part of the recorded versions), so the fixture holds
    digests only: tb_info, per-tensor gradient sums and L2 norms, per-tensor parameter sums after the
    update -- a 2.4M-parameter container per net would not be a small fixture."""
    from oracle.dsact_oracle_cnn import cnn_config, synth_image_batch

    obs_shape, A, conv_type, B, steps = (3, 96, 96), 3, "type_2", 8, 3
    kw = ref_loader.reference_kwargs(obs_shape, A, (256, 256, 256), act_limit=1.0)
    for key in ("value", "policy"):
        kw[key + "_func_type"] = "CNN"
        kw[key + "_conv_type"] = conv_type
        kw.pop(key + "_hidden_sizes")
    torch.manual_seed(0)
    alg = ref.DSAC_V2(**kw)
    nets = alg.networks
    cfg = cnn_config(obs_shape, A, conv_type)
    out = {"cfg_obs_shape": np.array(obs_shape), "cfg_act_dim": A, "cfg_conv_type": conv_type, "cfg_batch": B,
           "cfg_steps": steps, "versions": np.array([torch.__version__, np.__version__]),
           "keys": np.array(list(nets.state_dict().keys())),
           "shapes": np.array([str(list(v.shape)) for v in nets.state_dict().values()]),
           "init_sums": np.array([float(v.double().sum()) for v in nets.state_dict().values()])}
    for it in range(steps):
        d = synth_image_batch(cfg, B, seed=it)
        torch.manual_seed(1000 + it)
        tb = alg.local_update({k: v.clone() for k, v in d.items()}, it)
        out["s%d/tb" % it] = np.array([float(tb[k]) for k in TB_KEYS[:-1]], np.float64)
        online = list(nets.q1.parameters()) + list(nets.q2.parameters()) + list(nets.policy.parameters())
        out["s%d/grad_sum" % it] = np.array([float(p.grad.double().sum()) for p in online])
        out["s%d/grad_l2" % it] = np.array([float(p.grad.double().norm()) for p in online])
        out["s%d/param_sums" % it] = np.array([float(v.double().sum()) for v in nets.state_dict().values()])
    np.savez_compressed(os.path.join(out_dir, "step_cnn_type2.npz"), **out)
    print("wrote step_cnn_type2.npz")


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    ref = ref_loader.import_reference()
    torch.set_num_threads(1)
    for name, spec in CASES.items():
        gen_step_case(ref, name, spec, out_dir)
    gen_replay(out_dir)
    gen_checkpoint_layout(ref, out_dir)
    gen_cnn_case(ref, out_dir)
    gen_v1_case(out_dir)
    gen_humanoid_digest(ref, out_dir)


if __name__ == "__main__":
    main()
