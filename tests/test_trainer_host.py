"""Host-side loop logic of the trainer / sampler plugins on CPU (stub algorithm: HIP kernels need a GPU)."""
import json
import os
import sys

import numpy as np
import torch

from helpers import hip_kwargs

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs"))


class StubAlg:
    def __init__(self, networks):
        self.networks = networks
        self.calls = []

    def local_update(self, data, iteration):
        self.calls.append(iteration)
        return {"Loss/Critic loss-RL iter": 1.0 / (1 + iteration), "Time/Algorithm time [ms]-RL iter": 0.1}


class HostBuffer:
    """reference-style host ring used only to drive the trainer on CPU"""

    def __init__(self, O, A, N):
        from oracle.dsact_oracle import ReplayOracle
        self.r = ReplayOracle(O, A, N)

    size = property(lambda self: self.r.size)

    def add_batch(self, s):
        self.r.add_batch(s)

    def sample_batch(self, B):
        return self.r.sample_batch(B)

    def __get_RAM__(self):
        return 0.0


def test_sampler_and_trainer_loop(tmp_path):
    from dsac_v2_hip import ApproxContainer
    from plugin import create_evaluator, create_sampler, create_trainer
    kw = hip_kwargs(3, 1, (32, 32), 16, act_limit=2.0, env_id="synth_pendulum", sample_batch_size=20,
                    reward_scale=1, buffer_warm_size=100, max_iteration=12, log_save_interval=4,
                    apprfunc_save_interval=6, eval_interval=6, num_eval_episode=1, ini_network_dir=None,
                    save_folder=str(tmp_path), seed=3)
    torch.manual_seed(0)
    nets = ApproxContainer(**kw)  # CPU container: torch forward (acting path, not the update)
    sampler = create_sampler(**kw)
    sampler.networks = nets
    np.random.seed(0)
    torch.manual_seed(1)
    samples, tb = sampler.sample()
    assert len(samples) == 20 and len(samples[0]) == 8
    obs, info, act, rew, obs2, done, logp, info2 = samples[0]
    assert obs.shape == (3,) and act.shape == (1,) and abs(act[0]) <= 2.0 and done is False
    assert "Time/Sampler time [ms]-RL iter" in tb
    assert sampler.get_total_sample_number() == 20
    # 200-step time limit: truncation is stored as non-terminal and triggers a reset
    for _ in range(10):
        s, _ = sampler.sample()
        assert all(x[5] is False for x in s)
    alg = StubAlg(nets)
    trainer = create_trainer(alg, sampler, HostBuffer(3, 1, 1000), create_evaluator(**kw), **kw)
    assert trainer.buffer.size >= 100  # warm-up (reference trainer.py:50-52)
    trainer.train()
    assert alg.calls == list(range(12))
    files = sorted(os.listdir(tmp_path / "apprfunc"))
    assert "apprfunc_0.pkl" in files and "apprfunc_6.pkl" in files and "apprfunc_12.pkl" in files
    sd = torch.load(tmp_path / "apprfunc" / "apprfunc_12.pkl")
    assert list(sd.keys())[0] == "log_alpha" and "policy.policy.0.weight" in sd
    tags = {json.loads(l)["tag"] for l in open(tmp_path / "scalars.jsonl")}
    assert "Loss/Critic loss-RL iter" in tags and "Evaluation/1. TAR-RL iter" in tags
    # the reference opens the log with alg/sampler time 0 at step 0 (trainer.py:43-47) and its scripts export CSVs
    first = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")][:2]
    assert [r["tag"] for r in first] == ["Time/Algorithm time [ms]-RL iter", "Time/Sampler time [ms]-RL iter"]
    assert all(r["step"] == 0 and r["value"] == 0 for r in first)
    csvs = sorted(os.listdir(tmp_path / "data"))
    assert "Loss_Critic loss-RL iter.csv" in csvs and "Evaluation_1. TAR-RL iter.csv" in csvs
    assert open(tmp_path / "data" / "Loss_Critic loss-RL iter.csv").readline() == "Step,Value\n"
