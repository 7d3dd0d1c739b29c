"""Row a4 (OffSerialTrainer.step / train, training/trainer.py:60-152): the training LOOP against the unmodified reference.

  * CPU, where /root/reference is mounted: the reference's own algorithm and replay buffer are plugged into
    HipOffSerialTrainer + HipOffSampler + HipEvaluator and the run is compared with the reference's loop driven by its
    own factories (oracle/trainer_trajectory.py::run_reference) -- same seed, same env. With the arithmetic identical
    on both sides the loop logic must agree EXACTLY: replay indices, ring size/ptr, the ordered (tag, step, value) list
    of every scalar written, checkpoint file names and their order, evaluation returns, every update's tb_info.
  * `-m gpu`: the whole HIP stack (DSAC_V2_HIP + HipReplayBuffer + HIP acting forward, strict_rng) against the committed
    trajectory tests/golden/trainer_trajectory.json the reference produced: indices / cadence / file names exactly,
    statistics and evaluation returns at the parity gates of DESIGN section 5.
"""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import ref_loader
from oracle.trainer_trajectory import (ENVS, GOLDEN, TIME_TAGS, TRAINER_CASE, VARIANTS, Hooks, run_reference, tb_floats, variant_case,
                                       variant_golden)

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "dsac-v2_amd")
RAM_TAG = "RAM/RAM [MB]-RL iter"


def derived_kwargs(case, save_folder, **over):
    """what utils/init_args.py:11-83 adds to the argument dict (shapes and limits from the env, bookkeeping keys, the
    global seeds of utils/common_utils.py:140-157) -- restated here because init_args itself is outside the path and
    the GPU box has no reference checkout"""
    import plugin

    kw = dict(case, save_folder=save_folder, **over)
    env = plugin.create_env(**kw)
    kw["use_gpu"] = bool(kw.get("enable_cuda", False))
    kw["batch_size_per_sampler"] = kw["sample_batch_size"]
    shape = tuple(env.observation_space.shape)
    kw["obsv_dim"] = shape[0] if len(shape) == 1 else shape      # utils/init_args.py:29-33
    kw["action_dim"] = env.action_space.shape[0]
    kw["action_high_limit"] = env.action_space.high.astype("float32")
    kw["action_low_limit"] = env.action_space.low.astype("float32")
    kw["additional_info"] = {}
    kw["cnn_shared"] = False
    os.makedirs(os.path.join(save_folder, "apprfunc"), exist_ok=True)
    seed = int(kw["seed"])
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return kw


def run_hip_loop(kw, alg, buffer):
    """our sampler / evaluator / trainer around (alg, buffer); returns the same trajectory dict as run_reference"""
    import plugin

    sampler = plugin.create_sampler(**kw)
    evaluator = plugin.create_evaluator(**kw)
    updates, evals, buf_state, groups = [], [], [], []
    with Hooks() as hk:
        trainer = plugin.create_trainer(alg, sampler, buffer, evaluator, **kw)
        inner_update, inner_eval, inner_sample = alg.local_update, evaluator.run_evaluation, buffer.sample_batch

        def local_update(data, it):
            tb = inner_update(data, it)
            updates.append(tb_floats(tb))     # read at once: device statistics are kept for the last 16 updates only
            return tb

        alg.local_update = local_update
        if hasattr(alg, "local_update_group"):
            inner_group = alg.local_update_group

            def local_update_group(group, it):
                tb = inner_group(group, it)
                # a group's intermediate updates have no readable statistics (nobody logs them): recorded as None
                updates.extend([None] * (len(group) - 1) + [tb_floats(tb)])
                groups.append([int(it), len(group)])
                for _ in range(len(group)):
                    buf_state.append([int(buffer.size), int(buffer.ptr)])
                return tb

            alg.local_update_group = local_update_group
        evaluator.run_evaluation = lambda it: (lambda r: (evals.append([int(it), float(r)]), r)[1])(inner_eval(it))
        buffer.sample_batch = lambda n: (buf_state.append([int(buffer.size), int(buffer.ptr)]), inner_sample(n))[1]
        trainer.train()
    scalars = [[r["tag"], r["step"], r["value"]] for r in map(json.loads, open(os.path.join(kw["save_folder"], "scalars.jsonl")))]
    return {"indices": hk.indices, "buffer": buf_state, "scalars": scalars, "saved": hk.saved,
            "apprfunc_dir": sorted(os.listdir(os.path.join(kw["save_folder"], "apprfunc"))), "evals": evals,
            "tb_info": updates, "samples": int(sampler.get_total_sample_number()), "groups": groups}


def check_cadence(got, want):
    """everything about the loop that does not depend on floating point"""
    assert got["indices"] == want["indices"]
    assert got["buffer"] == want["buffer"]
    wall = "Evaluation/2. TAR-Total time [s]"      # its STEP is int(seconds since the trainer started) (trainer.py:125-129)
    key = lambda s: [s[0]] if s[0] == wall else s[:2]
    assert [key(s) for s in got["scalars"]] == [key(s) for s in want["scalars"]]
    assert got["saved"] == want["saved"] and got["apprfunc_dir"] == want["apprfunc_dir"]
    assert [e[0] for e in got["evals"]] == [e[0] for e in want["evals"]]
    assert got["samples"] == want["samples"] and len(got["tb_info"]) == len(want["tb_info"])


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")
def test_trainer_loop_equals_the_reference_loop_exactly(tmp_path):
    ref_loader.import_reference()
    for p in (PKG, ENVS):
        if p not in sys.path:
            sys.path.append(p)      # AFTER the reference root: `training`, `utils` stay the reference's packages
    import plugin

    plugin.install()
    want = run_reference(str(tmp_path / "ref"))
    # --- the same run with OUR loop around the reference's algorithm and buffer (arithmetic identical on both sides)
    from utils.initialization import create_alg, create_buffer

    kw = derived_kwargs(TRAINER_CASE, str(tmp_path / "hip"))
    alg = create_alg(**kw)
    # the reference builds sampler, buffer, evaluator in this order (example_train/main.py:160-168); run_hip_loop builds
    # the sampler and the evaluator (both draw a container from the torch generator), the buffer draws nothing
    buffer = create_buffer(**kw)
    got = run_hip_loop(kw, alg, buffer)
    check_cadence(got, want)
    assert got["tb_info"] == want["tb_info"]
    assert got["evals"] == want["evals"]
    for g, w in zip(got["scalars"], want["scalars"]):
        if g[0] in TIME_TAGS:
            continue
        assert g[2] == w[2] or (g[0] == RAM_TAG), (g, w)
    # the trainer's post-training CSV export exists for every tag the reference wrote
    tags = {s[0] for s in want["scalars"]}
    assert len(os.listdir(tmp_path / "hip" / "data")) == len(tags)


class _GroupedRefAlg:
    """the reference's algorithm behind the GROUP surface of DSAC_V2_HIP (local_update_group): K updates issued by one
    call. With identical arithmetic on both sides, HipOffSerialTrainer's grouping (which iterations it batches, when it
    draws their indices, where host-side events cut a group) must reproduce the reference loop exactly."""

    def __init__(self, alg):
        self._alg = alg
        self.networks = alg.networks

    def __getattr__(self, k):
        return getattr(self._alg, k)

    def local_update(self, data, it):
        return self._alg.local_update(data, it)

    def local_update_group(self, group, it):
        tb = None
        for j, batch in enumerate(group):
            tb = self._alg.local_update(batch, it + j)
        return tb


class _GroupedRefBuffer:
    def __init__(self, buf):
        self._buf = buf

    def __getattr__(self, k):
        return getattr(self._buf, k)

    def sample_batch(self, n):
        return self._buf.sample_batch(n)

    def sample_batches(self, batch_size, n):
        return [self._buf.sample_batch(batch_size) for _ in range(n)]


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not mounted")
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_grouped_updates_follow_the_reference_loop_exactly(tmp_path, variant):
    """sample_interval = K (training/trainer.py:63-66): HipOffSerialTrainer.train() issues the updates between two sampler
    calls as groups (sample_batches + local_update_group) cut at every iteration with a log / evaluation / checkpoint. Around
    the reference's own arithmetic that must be the reference loop to the last bit: index draws, ring state, every scalar,
    checkpoint names, evaluation returns, and the tb_info of every update a group can report (its last)."""
    ref_loader.import_reference()
    for p in (PKG, ENVS):
        if p not in sys.path:
            sys.path.append(p)
    import plugin

    plugin.install()
    case = variant_case(variant)
    want = run_reference(str(tmp_path / "ref"), case)
    from utils.initialization import create_alg, create_buffer

    kw = derived_kwargs(case, str(tmp_path / "hip"))
    alg = _GroupedRefAlg(create_alg(**kw))
    buffer = _GroupedRefBuffer(create_buffer(**kw))
    got = run_hip_loop(kw, alg, buffer)
    assert got["groups"], "no group was issued"
    K = case["sample_interval"]
    sizes = sorted({n for _, n in got["groups"]})
    assert sizes[-1] <= K and sizes[0] >= 2
    if variant == "si2":
        assert sizes == [2]
    if variant == "si8":
        assert len(sizes) >= 3          # the dense log / eval / save cadence cuts the groups of 8 into several lengths
    if variant == "si8_sparse":
        assert 8 in sizes               # whole groups of 8 between events
    # indices: the reference trainer draws per iteration, ours per group -- the same calls in the same order
    check_cadence(got, want)
    for g, w in zip(got["tb_info"], want["tb_info"]):
        assert g is None or g == w
    assert sum(g is not None for g in got["tb_info"]) >= len(want["tb_info"]) // K
    assert got["evals"] == want["evals"]
    for g, w in zip(got["scalars"], want["scalars"]):
        if g[0] in TIME_TAGS:
            continue
        assert g[2] == w[2] or (g[0] == RAM_TAG), (g, w)


def test_committed_trajectory_is_what_the_reference_produces_here(tmp_path):
    """the fixtures the GPU tests compare against are regenerated where the reference is mounted"""
    if not ref_loader.reference_available():
        pytest.skip("reference not mounted")
    for path, case in [(GOLDEN, None)] + [(variant_golden(v), variant_case(v)) for v in sorted(VARIANTS)]:
        want = json.load(open(path))
        got = run_reference(str(tmp_path / os.path.basename(path)), case)
        check_cadence(got, want)
        assert got["tb_info"] == want["tb_info"] and got["evals"] == want["evals"]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [None] + sorted(VARIANTS))
def test_hip_stack_follows_the_reference_trajectory(tmp_path, variant):
    """variant None: sample_interval 1 (one eager update per iteration). si2 / si8 / si8_sparse: the updates between two sampler
    calls run as graph replays (dsact_run_group: the pipelined graph, the reference's torch.randn draws through the noise
    table) -- against the trajectory the UNMODIFIED reference loop produced with that sample_interval."""
    for p in (PKG, ENVS):
        if p not in sys.path:
            sys.path.append(p)
    import plugin

    want = json.load(open(GOLDEN if variant is None else variant_golden(variant)))
    case = dict(want["case"], algorithm="DSAC_V2_HIP", buffer_name="hip_replay_buffer")
    kw = derived_kwargs(case, str(tmp_path), strict_rng=True)
    alg = plugin.create_alg(**kw)
    buffer = plugin.create_buffer(**kw)
    assert buffer.engine is alg.engine
    got = run_hip_loop(kw, alg, buffer)
    check_cadence(got, want)
    if variant is not None:
        assert got["groups"] and alg.engine.debug_get("graph_noise_table") == 1.0
        if variant == "si8_sparse":
            assert any(n == 8 for _, n in got["groups"])
        # (round 6: observation widths that are no multiple of 4 -- Pendulum's 3 -- run the row-slice chains, so these groups are
        #  replays of the pipelined graph with the noise table; the depth / mlp_separated variants take the tile path's merged graph)
    crit = 7   # Loss/Critic loss: a sum of squared TD terms -> relative gate
    for it, (g, w) in enumerate(zip(got["tb_info"], want["tb_info"])):
        if g is None:      # inside a group: not reported (the next reported update carries its effect)
            continue
        for k, (a, b) in enumerate(zip(g, w)):
            tol = 1e-6 + 1e-5 * abs(b) if k == crit else 1e-4
            assert abs(a - b) <= tol, (it, k, a, b)
    # evaluation: mean over 2 episodes of a sum of 60 rewards of magnitude <= 16 computed in the env's float64; the only
    # difference is the HIP acting forward (logits within 2e-5 of the CPU module) -> 1e-4 relative
    for (i0, a), (i1, b) in zip(got["evals"], want["evals"]):
        assert i0 == i1 and abs(a - b) <= 1e-4 * abs(b), (i0, a, b)
    vals = {}
    for g, w in zip(got["scalars"], want["scalars"]):
        if g[0] in TIME_TAGS or g[0] == RAM_TAG:
            continue
        tol = 1e-4 * max(1.0, abs(w[2]))
        assert abs(g[2] - w[2]) <= tol, (g, w)
        vals[g[0]] = g[2]
    assert len(vals) >= 15
    # checkpoints load back into a reference-shaped container (state_dict keys of dsac_v2.py:19-62)
    sd = torch.load(os.path.join(str(tmp_path), "apprfunc", "apprfunc_%d.pkl" % case["max_iteration"]))
    std_param = case.get("policy_std_type", "mlp_shared") == "parameter"
    if case.get("value_func_type") == "CNN":   # 173 keys, <net>.conv.* then the twin <net>.mean.* / <net>.log_std.* MLPs
        assert list(sd.keys())[0] == "log_alpha" and len(sd) == 173 and "policy.conv.0.weight" in sd and "q1_target.log_std.6.bias" in sd
        assert any(n == 8 for _, n in got["groups"])          # whole groups of 8 updates: the CNN examples' own sample_interval
        return
    if case.get("policy_std_type", "mlp_shared") == "mlp_separated":   # networks/mlp.py:46-57: two MLPs per policy net, 12 more tensors each
        assert list(sd.keys())[0] == "log_alpha" and len(sd) == 41 + 2 * 6 and "policy.policy.0.weight" not in sd
        assert tuple(sd["policy.log_std.4.weight"].shape) == (kw["action_dim"], kw["policy_hidden_sizes"][-1]) and "policy_target.mean.0.bias" in sd
        return
    if variant == "ragged_si2":      # (96, 40) / (40, 72): stored 128 wide, on the chains; the checkpoint holds the reference's shapes
        assert alg.engine.layout.pad_to == 128 and alg.engine.chain_active and alg.engine.debug_get("pipe_graph") == 1.0
        assert tuple(sd["q1.q.2.weight"].shape) == (40, 96) and tuple(sd["policy.policy.2.weight"].shape) == (72, 40) and len(sd) == 41
        return
    if len(kw["policy_hidden_sizes"]) != len(kw["value_hidden_sizes"]):   # each family with its own layer count
        n_lin = 4 * (len(kw["value_hidden_sizes"]) + 1) + 2 * (len(kw["policy_hidden_sizes"]) + 1)
        assert list(sd.keys())[0] == "log_alpha" and len(sd) == 1 + 4 + 2 * n_lin
        assert tuple(sd["policy.policy.%d.weight" % (2 * len(kw["policy_hidden_sizes"]))].shape) == (2 * kw["action_dim"], kw["policy_hidden_sizes"][-1])
        return
    assert list(sd.keys())[0] == "log_alpha" and len(sd) == (43 if std_param else 41)
    if std_param:   # the reference's own names for this policy_std_type (networks/mlp.py:63-73)
        assert tuple(sd["policy.log_std"].shape) == (1, kw["action_dim"]) and "policy.mean.0.weight" in sd and "policy.policy.0.weight" not in sd
