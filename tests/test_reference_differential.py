"""Differential tests against the LIVE reference: the unmodified reference's own
factories and sampler, run in this process through oracle/ref_loader.py, side by side with the HIP host layer.
Skipped where /root/reference is absent (the GPU box); no GPU is needed -- the acting path of a CPU container is torch.

  * reference `OffSampler(algorithm="DSAC_V2_HIP")` finds `dsac_v2_hip.ApproxContainer` by the reference's own
    discovery rule (training/off_sampler.py:19-23) and produces, transition for transition, what `HipOffSampler`
    produces from the same seeds (training/off_sampler.py:38-97)
  * reference `create_alg` / `create_buffer` (utils/initialization.py:48-110) resolve the HIP module and class names;
    without a GPU construction fails LOUDLY in the HIP layer (no silent CPU fallback)
  * the reference `ReplayBuffer` and the host mirror of the HIP ring agree on ptr/size/sample indices
"""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import hip_kwargs
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "dsac-v2_amd")


@pytest.fixture(scope="module")
def ref():
    mod = ref_loader.import_reference()
    for p in (PKG, os.path.join(HERE, "envs")):
        if p not in sys.path:
            sys.path.append(p)     # AFTER the reference root: `training`, `utils` stay the reference's packages
    import plugin

    plugin.install()
    return mod


def sampler_kwargs(**over):
    kw = hip_kwargs(3, 1, (32, 32), 16, act_limit=2.0, env_id="synth_pendulum", sample_batch_size=25,
                    batch_size_per_sampler=25, noise_params=None, reward_scale=1, seed=11, max_episode_steps=60)
    kw.update(over)
    return kw


def test_reference_sampler_with_hip_container_equals_hip_sampler(ref):
    import training.off_sampler as ref_sampler_mod          # the reference's
    from training.hip_sampler import HipOffSampler          # ours, found inside the same package after plugin.install()
    from utils.initialization import create_env as ref_create_env

    assert ref_sampler_mod.__file__.startswith(ref_loader.REFERENCE_ROOT)
    kw = sampler_kwargs()
    torch.manual_seed(5)
    ref_s = ref_sampler_mod.OffSampler(**kw)                # __import__("dsac_v2_hip").ApproxContainer(**kw)
    import dsac_v2_hip
    assert type(ref_s.networks) is dsac_v2_hip.ApproxContainer
    env = ref_create_env(**kw)                              # the same wrapper stack the reference sampler acts in
    env.seed(kw["seed"])
    hip_s = HipOffSampler(env=env, **kw)
    hip_s.networks = dsac_v2_hip.ApproxContainer(**kw)
    hip_s.load_state_dict(ref_s.networks.state_dict())
    assert np.array_equal(ref_s.obs, hip_s.obs)
    n_trunc = 0
    for call in range(6):                                   # 150 transitions: two 60-step time limits are crossed
        torch.manual_seed(100 + call)
        a, tb_a = ref_s.sample()
        torch.manual_seed(100 + call)
        b, tb_b = hip_s.sample()
        assert list(tb_a.keys()) == list(tb_b.keys()) == ["Time/Sampler time [ms]-RL iter"]
        assert len(a) == len(b) == kw["sample_batch_size"]
        for ta, tb_ in zip(a, b):
            obs_a, info_a, act_a, rew_a, obs2_a, done_a, logp_a, info2_a = ta
            obs_b, info_b, act_b, rew_b, obs2_b, done_b, logp_b, info2_b = tb_
            assert np.array_equal(obs_a, obs_b) and np.array_equal(obs2_a, obs2_b)
            assert np.array_equal(act_a, act_b) and act_a.dtype == act_b.dtype
            assert float(rew_a) == float(rew_b) and bool(done_a) == bool(done_b)
            assert np.array_equal(logp_a, logp_b)
            assert bool(info2_a["TimeLimit.truncated"]) == bool(info2_b["TimeLimit.truncated"])
            n_trunc += bool(info2_a["TimeLimit.truncated"])
    assert n_trunc == 2
    assert ref_s.get_total_sample_number() == hip_s.get_total_sample_number() == 150


def test_reference_factories_resolve_the_hip_modules(ref):
    from utils.initialization import create_alg, create_buffer
    from dsact._ffi import DsactError

    kw = sampler_kwargs(buffer_max_size=256)
    # the reference's rule: module algorithm.lower(), class algorithm -- reaches DSAC_V2_HIP.__init__, which refuses to
    # run without its GPU library / device instead of falling back to a CPU update
    with pytest.raises((DsactError, RuntimeError)) as ei:
        create_alg(**kw)
    assert "DSAC_V2_HIP" in "".join(str(f) for f in ei.traceback) or "dsac_v2_hip" in "".join(str(f.path) for f in ei.traceback)
    # "training." + buffer_name.lower(), class CamelCase(buffer_name): the module is found INSIDE the reference's package
    import importlib
    mod = importlib.import_module("training." + kw["buffer_name"])
    assert mod.__file__.startswith(PKG) and hasattr(mod, "HipReplayBuffer")
    import training.replay_buffer as ref_rb
    assert ref_rb.__file__.startswith(ref_loader.REFERENCE_ROOT)       # the reference's own buffer is still there
    with pytest.raises((DsactError, RuntimeError)) as ei:
        create_buffer(**kw)
    assert "hip_replay_buffer" in "".join(str(f.path) for f in ei.traceback)
    # DSAC_V1_HIP by the same rule
    with pytest.raises((DsactError, RuntimeError)):
        create_alg(**dict(kw, algorithm="DSAC_V1_HIP", TD_bound=10))


def test_reference_replay_buffer_vs_host_mirror(ref):
    """reference ReplayBuffer (training/replay_buffer.py:20-90) vs oracle ReplayOracle, which the GPU tests compare the
    HIP ring against bit for bit (tests/test_hip_parity.py::test_replay_ring_and_gather_bit_exact): ptr / size after
    every add_batch, and the sampled rows after the same np.random.seed."""
    import training.replay_buffer as ref_rb
    from oracle.dsact_oracle import ReplayOracle

    O, A, N = 5, 2, 37
    kw = dict(trainer="off_serial_trainer", seed=0, obsv_dim=O, action_dim=A, buffer_max_size=N, additional_info={})
    a, b = ref_rb.ReplayBuffer(**kw), ReplayOracle(O, A, N)
    rng = np.random.default_rng(4)
    for n in (1, 10, 26, 3, 37, 5):      # fills, wraps, a batch as large as the ring
        samples = [(rng.standard_normal(O).astype(np.float32), {}, rng.uniform(-1, 1, A).astype(np.float32),
                    float(rng.standard_normal()), rng.standard_normal(O).astype(np.float32), bool(rng.random() < .2),
                    np.float32(rng.standard_normal()), {}) for _ in range(n)]
        a.add_batch(samples)
        b.add_batch(samples)
        assert (a.size, a.ptr) == (b.size, b.ptr)
        np.random.seed(n)
        sa = a.sample_batch(16)
        np.random.seed(n)
        sb = b.sample_batch(16)
        for k in ("obs", "obs2", "act", "rew", "done", "logp"):
            assert torch.equal(sa[k], sb[k]), k


def test_csv_export_equals_the_reference_export(ref, tmp_path):
    """f3: after training the reference's scripts call save_tb_to_csv(save_folder) (utils/tensorboard_setup.py:121-139).
    The HIP trainer logs to scalars.jsonl; its export must produce the same files with the same bytes as the
    reference's own function fed with the same scalars (its event-file reader swapped for ours -- tensorboard is
    not installed here), and the tag dictionary must be the reference's."""
    import shutil

    import utils.tensorboard_setup as ref_tb
    from training.hip_trainer import TB_TAGS, _Scalars, read_scalars, save_tb_to_csv

    assert TB_TAGS == ref_tb.tb_tags
    ours, theirs = tmp_path / "ours", tmp_path / "theirs"
    ours.mkdir(); theirs.mkdir()
    w = _Scalars(str(ours))
    rng = np.random.default_rng(0)
    w.add_dict({TB_TAGS["alg_time"]: 0, TB_TAGS["sampler_time"]: 0}, 0)
    for it in range(0, 50, 10):
        w.add_dict({TB_TAGS["loss_critic"]: float(rng.standard_normal()) * 3.7, TB_TAGS["loss_actor"]: float(rng.standard_normal()),
                    "DSAC2/mean_std1": torch.tensor(0.1 * it + 1e-3), TB_TAGS["alg_time"]: 0.0712}, it)
        w.add(TB_TAGS["TAR of total time"], -1234.5678, it // 3)
        w.add(TB_TAGS["Buffer RAM of RL iteration"], 3092.0, it)
    w.flush()
    files = save_tb_to_csv(str(ours))
    shutil.copy(ours / "scalars.jsonl", theirs / "scalars.jsonl")
    orig = ref_tb.read_tensorboard
    ref_tb.read_tensorboard = read_scalars          # same {tag: {"x", "y"}} contract (tensorboard_setup.py:14-36)
    try:
        ref_tb.save_tb_to_csv(str(theirs))          # the reference's naming + pandas formatting, unmodified
    finally:
        ref_tb.read_tensorboard = orig
    names = sorted(os.listdir(theirs / "data"))
    assert names == sorted(os.path.basename(f) for f in files) and len(names) == 7
    assert "Time_Algorithm time [ms]-RL iter.csv" in names and "Evaluation_2. TAR-Total time [s].csv" in names
    for n in names:
        assert (ours / "data" / n).read_bytes() == (theirs / "data" / n).read_bytes(), n


def test_reference_evaluator_with_hip_container_equals_hip_evaluator(ref, tmp_path):
    """f3: the reference `Evaluator(algorithm="DSAC_V2_HIP")` (training/evaluator.py:9-84: mode() actions, episode
    return = sum of rewards, mean over num_eval_episode) against `HipEvaluator` on the same seeded environment and the
    same weights: identical evaluation returns, episode after episode."""
    import training.evaluator as ref_eval_mod
    from training.hip_trainer import HipEvaluator
    from utils.initialization import create_env as ref_create_env
    import dsac_v2_hip

    kw = sampler_kwargs(num_eval_episode=3, is_render=False, save_folder=str(tmp_path), eval_save=False, max_episode_steps=40)
    torch.manual_seed(9)
    ref_e = ref_eval_mod.Evaluator(**dict(kw))
    assert type(ref_e.networks) is dsac_v2_hip.ApproxContainer
    env = ref_create_env(**dict(kw, reward_scale=None, repeat_num=None))     # what the reference evaluator builds (:12-14)
    hip_e = HipEvaluator(eval_env=env, **kw)
    hip_e.networks = dsac_v2_hip.ApproxContainer(**kw)
    hip_e.networks.load_state_dict(ref_e.networks.state_dict())
    for it in range(3):
        a, b = ref_e.run_evaluation(it), hip_e.run_evaluation(it)
        assert float(a) == float(b) and np.isfinite(a), (it, a, b)
        assert abs(float(a)) > 1.0          # a real return, not an empty episode


def test_cpu_container_action_distribution_equals_the_reference(ref):
    """VERDICT r2 (row a5): `dsac_v2_hip.TanhGaussDistribution` -- what a CPU `ApproxContainer` hands samplers and
    evaluators -- against the reference's class (utils/act_distribution_cls.py:21-79): sample / rsample / mode /
    log_prob / entropy bit-equal from the same logits, limits and generator state."""
    import dsac_v2_hip
    from utils.act_distribution_cls import TanhGaussDistribution as RefDist

    torch.manual_seed(3)
    A = 5
    mean = torch.randn(7, A) * 1.5
    std = torch.exp(torch.clamp(torch.randn(7, A), -3.0, 0.5))
    logits = torch.cat([mean, std], dim=-1)
    hi = torch.tensor([2.0, 0.4, 1.0, 3.0, 0.5])
    lo = torch.tensor([-2.0, -0.4, -0.5, 1.0, -0.5])
    ours, theirs = dsac_v2_hip.TanhGaussDistribution(logits), RefDist(logits)
    for d in (ours, theirs):
        d.act_high_lim, d.act_low_lim = hi, lo
    for name in ("sample", "rsample"):
        torch.manual_seed(17)
        a0, lp0 = getattr(theirs, name)()
        s0 = torch.get_rng_state()
        torch.manual_seed(17)
        a1, lp1 = getattr(ours, name)()
        assert torch.equal(a0, a1) and torch.equal(lp0, lp1), name
        assert torch.equal(s0, torch.get_rng_state()), name     # the same amount of the global stream is consumed
    assert torch.equal(theirs.mode(), ours.mode())
    act = theirs.mode() * 0.9 + 0.05 * (hi + lo)
    assert torch.equal(theirs.log_prob(act), ours.log_prob(act))
    assert torch.equal(theirs.entropy(), ours.entropy())
