"""TEST INFRASTRUCTURE ONLY -- trajectory of the UNMODIFIED reference training loop.

`run_reference()` drives the reference's own `create_env -> init_args -> create_alg -> create_sampler -> create_buffer
-> create_evaluator -> create_trainer -> train()` (example_train/main.py:156-170, training/trainer.py:15-158) in place,
on CPU, through the gym / tensorboard stubs of oracle/ref_loader.py and the synthetic Pendulum of tests/envs, and
records everything observable about the loop from OUTSIDE the reference's code (nothing under /root/reference is edited;
the hooks wrap `np.random.randint`, `torch.save`, the stub `SummaryWriter` and two bound methods of the objects the
reference's factories returned):

    indices      per iteration, the replay indices `ReplayBuffer.sample_batch` drew (training/replay_buffer.py:86)
    buffer       per iteration, (size, ptr) when the minibatch was drawn
    scalars      the ordered list of (tag, step, value) the trainer wrote (add_scalars / add_scalar, trainer.py:43-135)
    saved        the ordered list of checkpoint file names passed to torch.save + the final listing of apprfunc/
    evals        (iteration, total average return) of every evaluation
    tb_info      the 14 statistics of every update (dsac_v2.py:188-204)
    samples      sampler.get_total_sample_number() at the end

`python -m oracle.trainer_trajectory` writes tests/golden/trainer_trajectory.json from it; `tests/` replays the same
configuration through HipOffSerialTrainer (CPU differential test with the reference algorithm plugged in; `-m gpu`
test of the whole HIP stack against the committed file). Never imported by the product, bench.py or smoke().
"""
import json
import os
import sys

import numpy as np
import torch

from oracle import ref_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENVS = os.path.join(ROOT, "tests", "envs")
GOLDEN = os.path.join(ROOT, "tests", "golden", "trainer_trajectory.json")

TIME_TAGS = ("Time/Algorithm time [ms]-RL iter", "Time/Sampler time [ms]-RL iter", "Evaluation/2. TAR-Total time [s]")

# the flat argument dict of example_train/main.py:16-152 for a run that is over in seconds: every interval fires
# several times, the best-checkpoint rule (trainer.py:95-110) is exercised from iteration max_iteration/5 on, the ring
# (200 rows) wraps during the run, a 60-step time limit is crossed by sampler and evaluator
TRAINER_CASE = dict(
    env_id="synth_pendulum", algorithm="DSAC_V2", enable_cuda=False, seed=12345,
    reward_scale=1, action_type="continu", is_render=False, is_adversary=False,
    value_func_name="ActionValueDistri", value_func_type="MLP", value_hidden_sizes=[64, 64],
    value_hidden_activation="gelu", value_output_activation="linear", value_min_log_std=-8, value_max_log_std=8,
    policy_func_name="StochaPolicy", policy_func_type="MLP", policy_act_distribution="TanhGaussDistribution",
    policy_hidden_sizes=[64, 64], policy_hidden_activation="gelu", policy_output_activation="linear",
    policy_min_log_std=-20, policy_max_log_std=0.5,
    value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4,
    gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2, TD_bound=1, bound=True,
    trainer="off_serial_trainer", max_iteration=40, ini_network_dir=None,
    buffer_name="replay_buffer", buffer_warm_size=120, buffer_max_size=200, replay_batch_size=64, sample_interval=1,
    sampler_name="off_sampler", sample_batch_size=20, noise_params=None,
    evaluator_name="evaluator", num_eval_episode=2, eval_interval=10, eval_save=False,
    save_folder=None, apprfunc_save_interval=15, log_save_interval=4, max_episode_steps=60,
)


# sample_interval > 1 (training/trainer.py:63-66; the reference's CNN examples run 8, example_train/dsacv2_cnn_carracing_offasync.py:133):
# the sampler is called every K-th iteration and K updates follow each call. `si2` / `si8` keep the dense log / eval / save
# cadence above (every group of updates is cut by host-side events), `si8_sparse` has whole groups of 8 between events.
VARIANTS = {
    "si2": dict(sample_interval=2),
    "si8": dict(sample_interval=8),
    "si8_sparse": dict(sample_interval=8, max_iteration=48, log_save_interval=16, eval_interval=24, apprfunc_save_interval=40),
    # policy_std_type = "parameter" (networks/mlp.py:63-73; utils/common_utils.py:55 reads the kwarg): the whole loop -- sampler,
    # evaluator, checkpoints with `policy.log_std` / `policy.mean.*` -- with the learnable-parameter log-std, groups of two updates
    "std_param_si2": dict(sample_interval=2, policy_std_type="parameter"),
    # policy_std_type = "mlp_separated" (networks/mlp.py:46-57): `mean` and `log_std` from two MLPs -- checkpoints with
    # `policy.mean.*` / `policy.log_std.*`, sampler and evaluator acting through the twin-trunk arena layout (round 6)
    "std_sep_si2": dict(sample_interval=2, policy_std_type="mlp_separated"),
    # value_hidden_sizes / policy_hidden_sizes of different widths AND depth (utils/common_utils.py:59-62 reads the lists per key)
    "depth_si2": dict(sample_interval=2, policy_hidden_sizes=[48, 32, 40]),
    # ragged AND unequal widths of the same depth: DSAC_V2_HIP stores them zero-padded to 128 and runs the row-slice chains + the
    # pipelined graph (round 6) -- sampler, evaluator and checkpoints see the reference's shapes through windows of the arenas
    "ragged_si2": dict(sample_interval=2, value_hidden_sizes=[96, 40], policy_hidden_sizes=[40, 72]),
    # policy_act_distribution = "GaussDistribution" (utils/act_distribution_cls.py:82-115): sampler, evaluator (mode() clamps the
    # mean) and the update without tanh squashing
    "gauss_si2": dict(sample_interval=2, policy_act_distribution="GaussDistribution"),
    # value_output_activation / policy_output_activation = "tanh" (utils/common_utils.py:16-45 -> networks/mlp.py:15-20)
    "out_tanh_si2": dict(sample_interval=2, value_output_activation="tanh", policy_output_activation="tanh"),
    # the configuration the reference itself runs sample_interval = 8 with: the CNN examples (example_train/
    # dsacv2_cnn_carracing_offasync.py:133, sample_batch_size 8 at :140) -- conv type_2 over a (3,96,96) image env (tests/envs/
    # synth_blob_data.py), whole groups of 8 between host-side events, batch 16 (the smallest the twin-trunk chain units take)
    "cnn_si8": dict(env_id="synth_blob", value_func_type="CNN", policy_func_type="CNN", value_conv_type="type_2", policy_conv_type="type_2",
                    value_hidden_sizes=[256, 256, 256], policy_hidden_sizes=[256, 256, 256], sample_interval=8, sample_batch_size=8,
                    replay_batch_size=16, buffer_warm_size=24, buffer_max_size=48, max_iteration=32, log_save_interval=16,
                    eval_interval=16, apprfunc_save_interval=24, num_eval_episode=1, max_episode_steps=20),
}


def variant_case(name):
    return dict(TRAINER_CASE, **VARIANTS[name])


def variant_golden(name):
    return os.path.join(ROOT, "tests", "golden", "trainer_trajectory_%s.json" % name)


class Hooks:
    """records the loop from outside; usable around the reference trainer and around HipOffSerialTrainer alike"""

    def __init__(self):
        self.indices, self.saved, self.rand_calls = [], [], 0
        self._randint, self._save = np.random.randint, torch.save

    def __enter__(self):
        def randint(*a, **k):
            out = self._randint(*a, **k)
            arr = np.asarray(out).astype(np.int64)
            # (HipReplayBuffer.sample_batches draws the rows of a whole group with ONE call of shape (n, batch): the same stream, in
            #  the same order, as the reference's n calls -- recorded per minibatch like them)
            self.indices.extend(arr.tolist() if arr.ndim == 2 else [arr.tolist()])
            return out

        def save(obj, f, *a, **k):
            self.saved.append(os.path.basename(str(f)))
            return self._save(obj, f, *a, **k)

        np.random.randint, torch.save = randint, save
        return self

    def __exit__(self, *exc):
        np.random.randint, torch.save = self._randint, self._save


class RecordingWriter:
    """stand-in for torch.utils.tensorboard.SummaryWriter (tensorboard is not installed here)"""

    log = []

    def __init__(self, log_dir=None, flush_secs=20):
        self.log_dir = log_dir

    def add_scalar(self, tag, value, step=None):
        RecordingWriter.log.append([str(tag), int(step), float(value)])

    def flush(self):
        pass


def install_writer_stub():
    import types

    m = types.ModuleType("torch.utils.tensorboard")
    m.SummaryWriter = RecordingWriter
    m.__stub__ = True
    sys.modules["torch.utils.tensorboard"] = m
    RecordingWriter.log = []


def wrap_alg(alg, store):
    inner = alg.local_update

    def local_update(data, iteration):
        tb = inner(data, iteration)
        store.append((iteration, tb))
        return tb

    alg.local_update = local_update


def tb_floats(tb):
    from oracle.dsact_oracle import TB_KEYS

    return [float(tb[k]) for k in TB_KEYS[:-1]]


def run_reference(save_folder, case=None):
    """the reference loop, in place, on CPU -> the trajectory dict described in the module docstring"""
    ref_loader.import_reference()
    install_writer_stub()
    if ENVS not in sys.path:
        sys.path.append(ENVS)
    from training.evaluator import create_evaluator
    from training.off_sampler import create_sampler
    from training.trainer import create_trainer
    from utils.init_args import init_args
    from utils.initialization import create_alg, create_buffer, create_env

    args = dict(case or TRAINER_CASE, save_folder=save_folder)
    env = create_env(**args)
    args = init_args(env, **args)                 # seeds python / numpy / torch (utils/common_utils.py:140-157)
    alg = create_alg(**args)
    sampler = create_sampler(**args)
    buffer = create_buffer(**args)
    evaluator = create_evaluator(**args)
    updates, evals, buf_state = [], [], []
    with Hooks() as hk:
        trainer = create_trainer(alg, sampler, buffer, evaluator, **args)   # warm-up sampling happens in here
        wrap_alg(alg, updates)
        inner_eval, inner_sample = evaluator.run_evaluation, buffer.sample_batch
        evaluator.run_evaluation = lambda it: (lambda r: (evals.append([int(it), float(r)]), r)[1])(inner_eval(it))
        buffer.sample_batch = lambda n: (buf_state.append([int(buffer.size), int(buffer.ptr)]), inner_sample(n))[1]
        trainer.train()
    return {
        "case": {k: v for k, v in (case or TRAINER_CASE).items()},
        "versions": [torch.__version__, np.__version__],
        "indices": hk.indices, "buffer": buf_state, "scalars": list(RecordingWriter.log), "saved": hk.saved,
        "apprfunc_dir": sorted(os.listdir(os.path.join(save_folder, "apprfunc"))),
        "evals": evals, "tb_info": [tb_floats(tb) for _, tb in updates],
        "samples": int(sampler.get_total_sample_number()),
    }


def main():
    import tempfile

    only = sys.argv[1:]   # python -m oracle.trainer_trajectory [variant ...]: regenerate just those (wall-clock scalars differ per run)
    for name in [None] + sorted(VARIANTS):
        if only and name not in only:
            continue
        path = GOLDEN if name is None else variant_golden(name)
        with tempfile.TemporaryDirectory() as d:
            traj = run_reference(d, None if name is None else variant_case(name))
        with open(path, "w") as f:
            json.dump(traj, f)
        print("wrote %s: %d updates, %d scalars, %d checkpoints, evals %s" % (
            path, len(traj["tb_info"]), len(traj["scalars"]), len(traj["saved"]), traj["evals"]))


if __name__ == "__main__":
    main()
