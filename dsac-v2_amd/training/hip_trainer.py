"""HipOffSerialTrainer -- the reference's serial off-policy loop (training/trainer.py:15-158) without
the per-iteration device ping-pong and without per-step host syncs.

    step():  sampler.sample() -> buffer.add_batch()           every sample_interval iterations
             buffer.sample_batch(B) -> alg.local_update()     minibatch stays in HBM (HipBatch token)
               train(): the updates between two sampler calls (sample_interval = K: the reference's CNN examples run 8) are
               issued as GROUPS -- buffer.sample_batches(B, n) -> alg.local_update_group(): one (pipelined) hipGraph replay
               for n updates -- whenever algorithm and buffer offer those two methods. A group ends at every iteration the
               loop has to log, evaluate or save at, so everything the host reads sees exactly the reference's state.
             log (reads the lazily materialised tb_info)      every log_save_interval
             evaluator.run_evaluation()                       every eval_interval
             torch.save(networks.state_dict())                every apprfunc_save_interval

tensorboard is optional (not installed in every image): scalars always go to <save_folder>/scalars.jsonl,
and to a SummaryWriter too when one can be imported.
"""
import json
import os
import time

import torch

__all__ = ["HipOffSerialTrainer", "HipEvaluator", "create_trainer", "create_evaluator", "save_tb_to_csv", "TB_TAGS"]

TB_TAGS = {  # the reference's tag set, same keys and strings (utils/tensorboard_setup.py:142-153)
    "TAR of RL iteration": "Evaluation/1. TAR-RL iter",
    "TAR of total time": "Evaluation/2. TAR-Total time [s]",
    "TAR of collected samples": "Evaluation/3. TAR-Collected samples",
    "TAR of replay samples": "Evaluation/4. TAR-Replay samples",
    "Buffer RAM of RL iteration": "RAM/RAM [MB]-RL iter",
    "loss_actor": "Loss/Actor loss-RL iter",
    "loss_critic": "Loss/Critic loss-RL iter",
    "alg_time": "Time/Algorithm time [ms]-RL iter",
    "sampler_time": "Time/Sampler time [ms]-RL iter",
    "critic_avg_value": "Train/Critic avg value-RL iter",
}
TB = {"tar_iter": TB_TAGS["TAR of RL iteration"], "tar_time": TB_TAGS["TAR of total time"],
      "tar_samples": TB_TAGS["TAR of collected samples"], "tar_replay": TB_TAGS["TAR of replay samples"],
      "ram": TB_TAGS["Buffer RAM of RL iteration"]}


def read_scalars(path):
    """{tag: {"x": steps, "y": values}} of <path>/scalars.jsonl in first-seen tag order -- what the reference's
    read_tensorboard (utils/tensorboard_setup.py:14-36) returns from the event files. Values are rounded through
    float32 like a tensorboard scalar summary stores them."""
    import numpy as np

    out = {}
    fn = os.path.join(path, "scalars.jsonl")
    if not os.path.exists(fn):
        return out
    with open(fn) as f:
        for line in f:
            if not line.strip():
                continue
            r = json.loads(line)
            d = out.setdefault(r["tag"], {"x": [], "y": []})
            d["x"].append(int(r["step"]))
            d["y"].append(float(np.float32(r["value"])))
    return {k: {"x": np.array(v["x"]), "y": np.array(v["y"])} for k, v in out.items()}


def save_tb_to_csv(path):
    """The reference's post-training export (utils/tensorboard_setup.py:121-139, called by every example script after
    trainer.train()): one `<path>/data/<tag with / -> _>.csv` per scalar tag with the columns `Step,Value`. Reads the
    scalars.jsonl this trainer writes (tensorboard is not needed); returns the list of files written."""
    import csv

    files = []
    data = read_scalars(path)
    for tag, d in data.items():
        name = tag.replace("\\", "/").replace("/", "_")
        csv_dir = os.path.join(path, "data")
        os.makedirs(csv_dir, exist_ok=True)
        fn = os.path.join(csv_dir, "{}.csv".format(name))
        with open(fn, "w", newline="") as f:
            w = csv.writer(f, lineterminator="\n")
            w.writerow(["Step", "Value"])
            for x, y in zip(d["x"], d["y"]):
                w.writerow([int(x), repr(float(y))])
        files.append(fn)
    return files


class _Scalars:
    def __init__(self, folder):
        self.f = open(os.path.join(folder, "scalars.jsonl"), "a") if folder else None
        self.tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter  # needs the tensorboard package
            self.tb = SummaryWriter(log_dir=folder, flush_secs=20) if folder else None
        except Exception:
            self.tb = None

    def add(self, tag, value, step):
        v = float(value)
        if self.f:
            self.f.write(json.dumps({"tag": tag, "step": int(step), "value": v}) + "\n")
        if self.tb is not None:
            self.tb.add_scalar(tag, v, step)

    def add_dict(self, d, step):
        for k, v in d.items():
            self.add(k, v, step)

    def flush(self):
        if self.f:
            self.f.flush()
        if self.tb is not None:
            self.tb.flush()


class HipEvaluator:
    """Deterministic-mode rollouts (reference training/evaluator.py:34-84): mean episode return of
    `num_eval_episode` episodes acting with dist.mode()."""

    def __init__(self, index=0, **kwargs):
        from plugin import create_env

        self.env = kwargs.get("eval_env") or kwargs.get("env") or create_env(**kwargs)
        if kwargs.get("seed") is not None and hasattr(self.env, "seed"):
            self.env.seed(kwargs["seed"])  # reference evaluator.py:15: set_seed(..., env) seeds with the plain seed
        self.networks = kwargs.get("networks")
        if self.networks is None and "algorithm" in kwargs:   # evaluator.py:16-20: consumes the torch generator like the reference
            from training.hip_sampler import _container
            self.networks = _container(**kwargs)
        self.num_eval_episode = kwargs.get("num_eval_episode", 5)

    def run_an_episode(self):
        out = self.env.reset()
        obs = out[0] if isinstance(out, tuple) else out
        rewards, done = [], False
        while not done:
            with torch.no_grad():
                logits = self.networks.policy(torch.from_numpy(obs.astype("float32")[None]))
                act = self.networks.create_action_distributions(logits).mode()[0].cpu().numpy()
            obs, r, done, info = self.env.step(act)
            rewards.append(r)
            done = bool(done) or bool(info.get("TimeLimit.truncated", False))
        return sum(rewards)   # the reference's own reduction (evaluator.py:71): same value to the last bit

    def run_evaluation(self, iteration):
        import numpy as np

        return np.mean([self.run_an_episode() for _ in range(self.num_eval_episode)])   # evaluator.py:74-78


class HipOffSerialTrainer:
    def __init__(self, alg, sampler, buffer, evaluator, **kwargs):
        self.alg, self.sampler, self.buffer, self.evaluator = alg, sampler, buffer, evaluator
        self.networks = alg.networks
        if sampler is not None:
            sampler.networks = self.networks  # act with the live learner weights (trainer.py:24-26)
        if evaluator is not None:
            evaluator.networks = self.networks
        if kwargs.get("ini_network_dir") is not None:
            self.networks.load_state_dict(torch.load(kwargs["ini_network_dir"]))
        # additive (absent in the reference, whose checkpoints hold the networks only): optimiser sidecar next to every
        # apprfunc_{it}.pkl when `save_optimizer_state` is set, `ini_optimizer_dir` to resume from one
        self.save_optimizer_state = bool(kwargs.get("save_optimizer_state", False))
        if kwargs.get("ini_optimizer_dir") is not None:
            side = torch.load(kwargs["ini_optimizer_dir"])
            alg.load_optimizer_state_dict(side)
        self.replay_batch_size = kwargs["replay_batch_size"]
        self.max_iteration = kwargs["max_iteration"]
        self.sample_interval = kwargs.get("sample_interval", 1)
        self.log_save_interval = kwargs["log_save_interval"]
        self.apprfunc_save_interval = kwargs["apprfunc_save_interval"]
        self.eval_interval = kwargs["eval_interval"]
        self.save_folder = kwargs.get("save_folder")
        self.best_tar = -float("inf")
        self.iteration = 0
        self.last_tar = None
        # additive: `hip_group_updates` (default True) -- see train(); False keeps one local_update call per iteration
        self.group_updates = bool(kwargs.get("hip_group_updates", True))
        self._grouping = False        # only while train() drives the loop (a caller of step() sees one update per call)
        self._seg_end, self._seg_tb = -1, None
        if self.save_folder:
            os.makedirs(os.path.join(self.save_folder, "apprfunc"), exist_ok=True)
        self.writer = _Scalars(self.save_folder)
        # the reference opens its log with these two at step 0 (training/trainer.py:43-47)
        self.writer.add_dict({TB_TAGS["alg_time"]: 0, TB_TAGS["sampler_time"]: 0}, 0)
        self.writer.flush()
        while sampler is not None and self.buffer.size < kwargs["buffer_warm_size"]:  # trainer.py:50-52
            samples, _ = sampler.sample()
            self.buffer.add_batch(samples)
        self.start_time = time.time()

    def _has_event(self, it):
        """the host reads device state at the end of iteration `it`: tb_info (log), the policy (evaluation), the networks (save)"""
        return (it % self.log_save_interval == 0 or (self.evaluator is not None and it % self.eval_interval == 0)
                or (self.save_folder and it % self.apprfunc_save_interval == 0))

    def _segment_end(self, it):
        """last iteration of the run of updates that may be issued at once from iteration `it` on: nothing but
        sample_batch + local_update happens between them in the reference loop (training/trainer.py:60-146) -- it stops before
        the next sampler call and at the first iteration with a host-side event; the last iteration of train() bounds it"""
        end = (it // self.sample_interval + 1) * self.sample_interval - 1 if self.sampler is not None else it + 63
        end = min(end, self.max_iteration - 1, it + 63)
        j = it
        while j < end and not self._has_event(j):
            j += 1
        return max(j, it)

    def _update(self, it):
        """the replay + learn part of iteration `it` (trainer.py:68-82); returns its tb_info, or None inside a group (whose
        updates were issued by the group's first iteration; only its LAST iteration has host-side events)"""
        if it <= self._seg_end:
            return self._seg_tb if it == self._seg_end else None
        n = 1
        if self._grouping and self.group_updates and hasattr(self.alg, "local_update_group") and hasattr(self.buffer, "sample_batches"):
            n = self._segment_end(it) - it + 1
        if n >= 2:
            group = self.buffer.sample_batches(self.replay_batch_size, n)
            self._seg_tb = self.alg.local_update_group(group, it)
            self._seg_end = it + n - 1
            return None
        batch = self.buffer.sample_batch(self.replay_batch_size)
        return self.alg.local_update(batch, it)

    def step(self):
        sampler_tb = {}
        if self.sampler is not None and self.iteration % self.sample_interval == 0:
            samples, sampler_tb = self.sampler.sample()
            self.buffer.add_batch(samples)
        alg_tb = self._update(self.iteration)
        if self.iteration % self.log_save_interval == 0:
            self.writer.add_dict(dict(alg_tb.items()), self.iteration)  # the only host sync of the update
            self.writer.add_dict(sampler_tb, self.iteration)
        if self.evaluator is not None and self.iteration % self.eval_interval == 0:
            tar = self.evaluator.run_evaluation(self.iteration)
            self.last_tar = tar
            if tar >= self.best_tar and self.iteration >= self.max_iteration / 5 and self.save_folder:
                self.best_tar = tar
                d = os.path.join(self.save_folder, "apprfunc")
                for fn in os.listdir(d):
                    if fn.endswith("_opt.pkl"):
                        os.remove(os.path.join(d, fn))
                torch.save(self.networks.state_dict(), os.path.join(d, "apprfunc_{}_opt.pkl".format(self.iteration)))
            self.writer.add(TB["ram"], self.buffer.__get_RAM__(), self.iteration)
            self.writer.add(TB["tar_iter"], tar, self.iteration)
            self.writer.add(TB["tar_replay"], tar, self.iteration * self.replay_batch_size)
            self.writer.add(TB["tar_time"], tar, int(time.time() - self.start_time))
            if self.sampler is not None:
                self.writer.add(TB["tar_samples"], tar, self.sampler.get_total_sample_number())
        if self.save_folder and self.iteration % self.apprfunc_save_interval == 0:
            self.save_apprfunc()

    def train(self):
        self._grouping = True
        try:
            while self.iteration < self.max_iteration:
                self.step()
                self.iteration += 1
        finally:
            self._grouping = False
        if self.save_folder:
            self.save_apprfunc()
        self.writer.flush()
        if self.save_folder:
            save_tb_to_csv(self.save_folder)   # what the reference's example scripts do right after train()

    def save_apprfunc(self):
        torch.save(self.networks.state_dict(),
                   os.path.join(self.save_folder, "apprfunc", "apprfunc_{}.pkl".format(self.iteration)))
        if self.save_optimizer_state and hasattr(self.alg, "optimizer_state_dict"):
            side = dict(self.alg.optimizer_state_dict(), iteration=int(self.iteration))
            torch.save(side, os.path.join(self.save_folder, "apprfunc", "apprfunc_{}.optstate.pkl".format(self.iteration)))


def create_trainer(alg, sampler, buffer, evaluator, **kwargs):
    return HipOffSerialTrainer(alg, sampler, buffer, evaluator, **kwargs)


def create_evaluator(**kwargs):
    return HipEvaluator(**kwargs)
