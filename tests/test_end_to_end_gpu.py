"""BASELINE.json configs[0] end to end on the HIP path: Pendulum dynamics, the full plugin stack
(create_alg / create_buffer / create_sampler / create_evaluator / create_trainer), MLP 3x256 GELU."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import hip_kwargs

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs"))
pytestmark = pytest.mark.gpu


def test_pendulum_end_to_end_learns(tmp_path):
    import plugin
    kw = hip_kwargs(3, 1, (256, 256, 256), 256, act_limit=2.0, env_id="synth_pendulum", sample_batch_size=20,
                    reward_scale=1, buffer_warm_size=1000, buffer_max_size=100000, max_iteration=9001,
                    log_save_interval=500, apprfunc_save_interval=4500, eval_interval=1500, num_eval_episode=5,
                    ini_network_dir=None, save_folder=str(tmp_path), seed=12345, sample_interval=1)
    torch.manual_seed(kw["seed"]); np.random.seed(kw["seed"])
    alg = plugin.create_alg(**kw)
    sampler = plugin.create_sampler(**kw)
    buf = plugin.create_buffer(**kw)
    assert buf.engine is alg.engine  # the minibatch never leaves HBM
    ev = plugin.create_evaluator(**kw)
    tr = plugin.create_trainer(alg, sampler, buf, ev, **kw)
    tars = []
    orig = ev.run_evaluation
    ev.run_evaluation = lambda it: tars.append(orig(it)) or tars[-1]
    tr.train()
    print("eval TAR per 1500 iterations:", [round(t, 1) for t in tars])
    assert len(tars) == 7 and all(np.isfinite(tars))
    # the shipped reference run (results/DSAC_V2_gym_pendulum): -1476 @0, -1104 @4k, -264 @6k, -122 @8k
    assert max(tars[3:]) > -900 and max(tars[3:]) > tars[0] + 300, tars
    st = alg.engine.read_stats()
    assert all(np.isfinite(v) for v in st.values())
    assert 0.05 < st["DSAC2/alpha-RL iter"] < 2.8       # alpha adapts downwards from e
    assert os.path.exists(tmp_path / "apprfunc" / "apprfunc_9000.pkl")
    assert alg.engine.get_state()["adam_steps"] == [9001, 4501, 4501]


def test_cnn_image_env_end_to_end_learns(tmp_path):
    """BASELINE.json configs[3] shape through the whole plugin stack: image observations (3,96,96), conv type_2
    approximators, image replay ring in HBM, acting through dsact_policy_forward. The reward depends on reading a
    position from the pixels, so improving over the initial policy exercises the conv forward AND backward."""
    import plugin
    from test_hip_cnn_parity import cnn_kwargs

    kw = cnn_kwargs((3, 96, 96), 3, "type_2", 64, env_id="synth_blob", sample_batch_size=20, reward_scale=1,
                    buffer_warm_size=400, buffer_max_size=50000, max_iteration=5501, log_save_interval=500,
                    apprfunc_save_interval=5500, eval_interval=500, num_eval_episode=5, ini_network_dir=None,
                    save_folder=str(tmp_path), seed=2024, sample_interval=1, strict_rng=False)
    torch.manual_seed(kw["seed"]); np.random.seed(kw["seed"])
    alg = plugin.create_alg(**kw)
    sampler = plugin.create_sampler(**kw)
    buf = plugin.create_buffer(**kw)
    assert buf.engine is alg.engine
    ev = plugin.create_evaluator(**kw)
    tr = plugin.create_trainer(alg, sampler, buf, ev, **kw)
    tars = []
    orig = ev.run_evaluation
    ev.run_evaluation = lambda it: tars.append(orig(it)) or tars[-1]
    tr.train()
    print("eval TAR per 500 iterations:", [round(t, 2) for t in tars])
    assert len(tars) == 12 and all(np.isfinite(tars))
    # 20 steps/episode; a policy blind to the image scores about -20*(1/3 + var(a0)) <= -6.7; reading it gets close
    # to 0. Observed: -7.4, -6.3, ... (entropy-dominated while alpha decays from e) ..., -2.2 @4.5k, -1.5 @5k
    assert max(tars[-3:]) > -4.0 and max(tars[-3:]) > tars[0] + 2.0, tars
    st = alg.engine.read_stats()
    assert all(np.isfinite(v) for v in st.values())
    sd = torch.load(tmp_path / "apprfunc" / "apprfunc_5500.pkl")
    assert len(sd) == 173 and tuple(sd["policy.conv.0.weight"].shape) == (8, 3, 4, 4)


def test_v1_pendulum_end_to_end_learns(tmp_path):
    """DSAC_V1_HIP (reference dsac_v1.py on the shared kernels) through the same plugin stack."""
    import plugin
    kw = hip_kwargs(3, 1, (256, 256, 256), 256, act_limit=2.0, env_id="synth_pendulum", sample_batch_size=20,
                    reward_scale=1, buffer_warm_size=1000, buffer_max_size=100000, max_iteration=9001,
                    log_save_interval=500, apprfunc_save_interval=4500, eval_interval=1500, num_eval_episode=5,
                    ini_network_dir=None, save_folder=str(tmp_path), seed=12345, sample_interval=1,
                    algorithm="DSAC_V1_HIP", TD_bound=10)
    torch.manual_seed(kw["seed"]); np.random.seed(kw["seed"])
    alg = plugin.create_alg(**kw)
    assert type(alg).__name__ == "DSAC_V1_HIP"
    sampler = plugin.create_sampler(**kw)
    buf = plugin.create_buffer(**kw)
    assert buf.engine is alg.engine
    ev = plugin.create_evaluator(**kw)
    tr = plugin.create_trainer(alg, sampler, buf, ev, **kw)
    tars = []
    orig = ev.run_evaluation
    ev.run_evaluation = lambda it: tars.append(orig(it)) or tars[-1]
    tr.train()
    print("DSAC_V1 eval TAR per 1500 iterations:", [round(t, 1) for t in tars])
    assert len(tars) == 7 and all(np.isfinite(tars))
    assert max(tars[3:]) > -900 and max(tars[3:]) > tars[0] + 300, tars
    sd = torch.load(tmp_path / "apprfunc" / "apprfunc_9000.pkl")
    assert list(sd.keys())[:2] == ["log_alpha", "q.q.0.weight"] and "q_target.q.0.weight" in sd
    assert alg.engine.get_state()["adam_steps"] == [9001, 4501, 4501]
