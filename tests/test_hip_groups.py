"""dsact_run_group -- the updates between two sampler calls of the reference loop (training/trainer.py:63-82 with
sample_interval = K) as ONE graph replay -- against the same updates issued one by one, bit for bit, and (strict RNG through
the noise table) against the oracle directly.

  * device Philox noise: group replays of several lengths (graphs are cached per length; the pipelined graph where the shape
    allows it) == { dsact_gather; dsact_step } per update: parameters, targets, both Adam moments, step state, statistics,
    the staged minibatch;
  * reference noise (`strict_rng`): the reference's torch.randn draws travel as the graph's noise table -- == eager strict
    updates bitwise, and the pipelined graph's results against the ORACLE fed the same minibatches and noise (no
    test fed reference noise through the pipelined graph itself);
  * the plugin surface: HipReplayBuffer.sample_batches + DSAC_V2_HIP.local_update_group == sample_batch + local_update.
"""
import numpy as np
import pytest
import torch

from helpers import hip_kwargs
from oracle.dsact_oracle import TB_KEYS, draw_noise
from test_hip_parity import Report, make_pair

pytestmark = pytest.mark.gpu


def host_ring(N, O, A, seed):
    rng = np.random.default_rng(seed)
    return {"obs": rng.standard_normal((N, O), dtype=np.float32), "act": rng.uniform(-0.4, 0.4, (N, A)).astype(np.float32),
            "rew": rng.standard_normal(N, dtype=np.float32), "obs2": rng.standard_normal((N, O), dtype=np.float32),
            "done": (rng.random(N) < 0.05).astype(np.float32)}


def fill(e, ring):
    e.buffer_create(ring["rew"].shape[0])
    e.buffer_add(ring["obs"], ring["act"], ring["rew"], ring["obs2"], ring["done"])
    e.sync()


def same_engine_state(a, b, tag):
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(a, name), getattr(b, name)), (tag, name)
    assert a.get_state() == b.get_state(), tag
    sa, sb = a.read_stats(), b.read_stats()
    for k in sa:
        if k.startswith("_device"):
            continue
        assert sa[k] == sb[k] or (np.isnan(sa[k]) and np.isnan(sb[k])), (tag, k, sa[k], sb[k])
    ba, bb = a.read_batch(with_logp=False), b.read_batch(with_logp=False)
    for k in ("obs", "act", "rew", "obs2", "done"):
        assert np.array_equal(ba[k], bb[k]), (tag, k)


@pytest.mark.parametrize("O,A,hid,B,D,first,lengths", [
    (16, 4, (64, 64), 64, 2, 0, [8, 8, 3, 1, 2, 8, 3]),          # lengths recur: cached graphs are re-activated
    (16, 4, (64, 64), 64, 2, 5, [7, 2, 7]),                       # starts on an odd iteration, odd lengths: both phase graphs
    (24, 6, (128, 128, 128), 32, 3, 1, [6, 4, 6]),                # delay_update 3
    (11, 3, (96, 40), 50, 2, 0, [4, 5]),                          # tile path (ragged widths): the plain merged graph
    (376, 17, (256, 256, 256), 256, 2, 3, [8, 8, 5]),             # the BASELINE.json shape
])
def test_run_group_equals_eager_steps(O, A, hid, B, D, first, lengths):
    N = 3000
    ring = host_ring(N, O, A, 3)
    engines = []
    for mode in ("eager", "group"):
        alg, _ = make_pair(O, A, hid, B, seed=4, delay_update=D)
        e = alg.engine
        e.set_device_rng(4242)
        fill(e, ring)
        np.random.seed(7)
        it = first
        for n in lengths:
            rows = np.stack([np.random.randint(0, N, size=B) for _ in range(n)])
            if mode == "group":
                e.run_group(it, rows)
            else:
                for j in range(n):
                    e.gather(rows[j])
                    e.step(it + j)
            it += n
        e.sync()
        engines.append(e)
    same_engine_state(engines[0], engines[1], "group vs eager")
    g = engines[1]
    assert g.debug_get("graph_cache") == len(set(lengths)) - 1       # one active graph, the other lengths cached
    if g.chain_active:
        assert g.debug_get("pipe_graph") == (1.0 if lengths[-1] >= 2 else 0.0)
    assert torch.isfinite(g.online).all()


def noise_row(nz):
    return np.concatenate([nz["eps_new"].numpy().reshape(-1), nz["eps_2"].numpy().reshape(-1), nz["z5"].numpy(), nz["z6"].numpy()])


@pytest.mark.parametrize("O,A,hid,B,first,lengths", [
    (16, 4, (64, 64), 64, 1, [4, 3, 4]),
    (376, 17, (256, 256, 256), 256, 0, [8]),                      # the BASELINE.json shape, one pipelined graph of 8 updates
])
def test_reference_noise_through_the_pipelined_graph(O, A, hid, B, first, lengths):
    """strict RNG: the reference's eight torch.randn draws per update are drawn up front, in its order,
    and replayed from the graph's noise table. (1) == the same updates issued eagerly with dsact_set_noise, bit for bit;
    (2) the graph's results against the ORACLE on the same minibatches and noise at the parity gates."""
    N = 2000
    ring = host_ring(N, O, A, 5)
    total = sum(lengths)
    np.random.seed(11)
    rows = np.stack([np.random.randint(0, N, size=B) for _ in range(total)])
    torch.manual_seed(99)
    noises = [draw_noise(B, A) for _ in range(total)]
    engines, orc = [], None
    for mode in ("eager", "group"):
        alg, o = make_pair(O, A, hid, B, seed=4)
        e = alg.engine
        fill(e, ring)
        k = 0
        for n in lengths:
            if mode == "group":
                e.run_group(first + k, rows[k:k + n], np.stack([noise_row(z) for z in noises[k:k + n]]))
            else:
                for j in range(k, k + n):
                    z = noises[j]
                    e.set_noise(z["eps_new"].numpy(), z["eps_2"].numpy(), z["z5"].numpy(), z["z6"].numpy())
                    e.gather(rows[j])
                    e.step(first + j)
            k += n
        e.sync()
        engines.append(e)
        orc = o
    same_engine_state(engines[0], engines[1], "noise table vs eager strict")
    g = engines[1]
    assert g.debug_get("graph_noise_table") == 1.0 and g.debug_get("pipe_graph") == 1.0
    # ---- the oracle on the same minibatches and noise
    tb = None
    for j in range(total):
        r = rows[j]
        data = {k: torch.as_tensor(ring[k][r]) for k in ("obs", "act", "rew", "obs2", "done")}
        data["logp"] = torch.zeros(B)
        tb = orc.local_update(data, noises[j], first + j)
    rep = Report("reference noise through the pipelined graph: O=%d A=%d hid=%s B=%d, %d updates vs oracle" % (O, A, hid, B, total))
    st = g.read_stats()
    for k in TB_KEYS[:-1]:
        if k == "Loss/Critic loss-RL iter":
            rep.cmp("tb." + k, st[k], float(tb[k]), 1e-6, 1e-5)
        else:
            rep.cmp("tb." + k, st[k], float(tb[k]), 1e-4)
    got, want = g.online.cpu().numpy().astype(np.float64), orc.flat_params().numpy().astype(np.float64)
    err = np.abs(got - want)
    # (the element-wise Adam-noise budget is enforced on the eager path by test_hip_parity.run_case; the graph equals that
    #  path bitwise -- here: every element within the sign-flip worst case, all but a few per mille within 1e-6)
    frac = float((err > 1e-6).mean())
    rep.rows.append(("params: fraction over 1e-6", frac, 0.0, 2e-3, frac <= 2e-3))
    worst_ok = float(err.max()) <= 2.0 * 1e-4 * total
    rep.rows.append(("params: worst element", float(err.max()), float(np.abs(want).max()), 2e-4 * total, worst_ok))
    if frac > 2e-3 or not worst_ok:
        rep.bad.append("params")
    rep.cmp("targets", g.target.cpu().numpy(), orc.flat_targets(), 1e-6)
    rep.finish()


def test_group_surface_equals_per_iteration_surface():
    """HipReplayBuffer.sample_batches + DSAC_V2_HIP.local_update_group (what HipOffSerialTrainer.train() issues between two
    sampler calls) == sample_batch + local_update per iteration: same NumPy stream consumption, same parameters, and the
    group's tb_info is the last update's."""
    from dsac_v2_hip import DSAC_V2_HIP
    from training.hip_replay_buffer import HipReplayBuffer

    O, A, hid, B, N, K = 16, 4, (64, 64), 64, 500, 6
    ring = host_ring(N, O, A, 9)
    out = []
    for mode in ("single", "group"):
        torch.manual_seed(2)
        kw = hip_kwargs(O, A, hid, B, buffer_max_size=N, seed=5)
        alg = DSAC_V2_HIP(**kw)
        buf = HipReplayBuffer(**kw)
        assert buf.engine is alg.engine
        samples = [(ring["obs"][i], {}, ring["act"][i], float(ring["rew"][i]), ring["obs2"][i], bool(ring["done"][i]), 0.0, {})
                   for i in range(N)]
        buf.add_batch(samples)
        np.random.seed(3)
        if mode == "group":
            g1 = buf.sample_batches(B, K)
            tb = alg.local_update_group(g1, 10)
            g2 = buf.sample_batches(B, 2)
            tb = alg.local_update_group(g2, 10 + K)
        else:
            for it in range(10, 10 + K + 2):
                tb = alg.local_update(buf.sample_batch(B), it)
        vals = [float(tb[k]) for k in TB_KEYS[:-1]]
        out.append((alg, vals, np.random.randint(0, 1 << 30)))
    assert out[0][2] == out[1][2]                       # the NumPy stream stands where the reference loop would leave it
    assert out[0][1] == out[1][1]
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(out[0][0].engine, name), getattr(out[1][0].engine, name)), name
    # a token whose rows were overwritten after sampling refuses to train (the reference's batch is a copy taken at sample time)
    alg, buf = out[1][0], None
    from dsac_v2_hip import HipBatchGroup
    grp = HipBatchGroup(alg.engine, np.zeros((2, B), np.int64))
    alg.engine.buffer_add(ring["obs"][:N], ring["act"][:N], ring["rew"][:N], ring["obs2"][:N], ring["done"][:N])
    with pytest.raises(RuntimeError, match="overwritten"):
        alg.local_update_group(grp, 0)


@pytest.mark.parametrize("O,A,hid,B,first,total", [(16, 4, (64, 64), 64, 0, 8), (376, 17, (256, 256, 256), 256, 2, 8)])
def test_fast_mode_takes_the_pipelined_graph(O, A, hid, B, first, total):
    """DSACT_F_SKIP_ACTOR_ON_OFF_ITERS ("fast": the discarded policy backward of the policy-preserving updates is not computed,
    dsac_v2.py:174-186 vs :324) now captures the pipelined graph too (fast was slower than strict because it fell
    back to the plain graph): == eager fast updates == the strict trajectory's parameters, bit for bit."""
    N = 2048
    engines = []
    for mode in ("eager_fast", "graph_fast", "graph_strict"):
        alg, _ = make_pair(O, A, hid, B, seed=4)
        e = alg.engine
        e.set_device_rng(31)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(8, B)))
        if mode == "eager_fast":
            e.time_steps(first, total, use_graph=False, flags=1)
        else:
            fl = 1 if mode == "graph_fast" else 0
            e.graph_build(4, fl)
            assert e.debug_get("pipe_graph") == 1.0
            e.graph_run(first, total)
        e.sync()
        engines.append(e)
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(engines[0], name), getattr(engines[1], name)), ("fast graph vs fast eager", name)
    for name in ("online", "target"):
        assert torch.equal(getattr(engines[1], name), getattr(engines[2], name)), ("fast vs strict", name)


def test_state_acknowledgement_is_explicit():
    """dsact_set_state acknowledges a hand-over timeout only as a FULL restore (step counters AND the EMA); a
    partial call leaves the handle refusing updates; dsact_debug_set("ack_state") is the explicit acknowledgement."""
    from dsact._ffi import DsactError

    O, A, hid, B = 16, 4, (64, 64), 64
    alg, _ = make_pair(O, A, hid, B, seed=4)
    e = alg.engine
    e.set_device_rng(5)
    ring = host_ring(512, O, A, 1)
    fill(e, ring)
    e.upload_index_table(np.random.default_rng(0).integers(0, 512, size=(4, B)))
    state = e.get_state()
    e.debug_set("withhold_flag", 1)
    e.graph_build(4)
    e.graph_run(0, 4)
    with pytest.raises(DsactError):
        e.sync()
    assert e.debug_get("state_invalid") == 1.0
    e.debug_set("withhold_flag", 0)
    e.set_state(mean_std=state["mean_std"])                 # partial: not an acknowledgement
    assert e.debug_get("state_invalid") == 1.0
    with pytest.raises(DsactError):
        e.step(0)
    e.debug_set("ack_state", 1)
    assert e.debug_get("state_invalid") == 0.0
    e.gather(np.arange(B))
    e.step(0)
    e.sync()


class _ToyEnv:     # deterministic toy dynamics with a time limit (gym 0.23 protocol)
    class _S:
        low, high = np.full(4, -0.3, np.float32), np.full(4, 0.3, np.float32)
    action_space = _S()

    def __init__(self):
        self.t, self.s = 0, np.zeros(16, np.float32)

    def reset(self):
        self.t, self.s = 0, np.linspace(-1, 1, 16).astype(np.float32)
        return self.s.copy(), {}

    def step(self, a):
        self.t += 1
        self.s = (0.9 * self.s + 0.1 * np.resize(a, 16)).astype(np.float32)
        return self.s.copy(), float(self.s.sum()), False, {"TimeLimit.truncated": self.t >= 7}


@pytest.mark.parametrize("hid,env,kw,fast", [
    ((64, 64), {}, {}, True),
    ((64, 64, 64, 64, 64), {}, {}, False),                      # 5 hidden layers: beyond the one-launch acting forward (kActMaxLayers)
    ((64, 64), {"DSACT_NO_FAST_ACT": "1"}, {}, False),          # the documented A/B switch
    ((64, 64), {}, {"hip_sampler_general_path": True}, False),  # forced by the caller
])
def test_sampler_fast_path_gate_is_the_librarys(hid, env, kw, fast, monkeypatch):
    """HipOffSampler took dsact_act_sample for every attached MLP policy with obs_dim <= 768, but the
    library also wants <= 4 hidden layers and DSACT_NO_FAST_ACT unset -- sample() aborted training on those. The gate is now
    the library's own answer (dsact_debug_get "act_fast"); every configuration samples."""
    from training.hip_sampler import HipOffSampler

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alg, _ = make_pair(16, 4, hid, 32, act_limit=0.4, seed=62)
    assert alg.engine.debug_get("act_fast") == (1.0 if fast or kw else 0.0)
    smp = HipOffSampler(env=_ToyEnv(), networks=alg.networks, sample_batch_size=12, action_type="continu", **kw)
    torch.manual_seed(9)
    batch, tb = smp.sample()
    assert len(batch) == 12 and (getattr(batch, "packed", None) is not None) == fast
    for s in batch:
        assert np.isfinite(np.asarray(s[2])).all() and np.all(np.abs(np.asarray(s[2])) <= 0.4 + 1e-6)


def test_merged_critic_backward_handover_poisoned_between_replays(monkeypatch):
    """k_chain_bwd_qt (round 5): the critics' chains hand their dZ packs / dL/dout to the critics' weight-gradient tiles of the
    SAME launch, layer by layer, with the arrival counter of layer l raised one layer late and no drain of the weight stream
    (dsact_chain.h: the `s_waitcnt vmcnt(kPD)` argument). Every handed-over buffer is filled with NaN before EVERY replay of a
    two-update graph whose first update runs the merged launch: a tile that passed its counter before the producers' stores
    had landed would compute on NaN. 80 replays at the BASELINE shape and 80 at a 64-wide one (one stream trip per layer):
    merged == two launches (DSACT_NO_BQT_MERGE=1) bit for bit, all finite, no hand-over failure."""
    for O, A, hid, B in ((376, 17, (256, 256, 256), 256), (16, 4, (64, 64), 64)):
        engines = []
        for merged in (True, False):
            if not merged:
                monkeypatch.setenv("DSACT_NO_BQT_MERGE", "1")
            alg, _ = make_pair(O, A, hid, B, seed=31)
            monkeypatch.delenv("DSACT_NO_BQT_MERGE", raising=False)
            e = alg.engine
            e.set_device_rng(77)
            N = 4096
            e.buffer_create(N)
            g = torch.Generator(device="cuda").manual_seed(3)
            e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                                 torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                                 (torch.rand(N, device="cuda", generator=g) < .05).float())
            np.random.seed(2)
            e.upload_index_table(np.random.randint(0, N, size=(8, B)))
            e.graph_build(2)
            names = [n for n, _, _ in e.profile_steps(1, 2)]
            assert ("chain_bwd_qt" in names) == merged, names
            for rep in range(80):
                e.debug_set("poison_handover", float("nan"))
                e.graph_run(3 + 2 * rep, 2)
            e.sync()
            engines.append(e)
        for name in ("online", "target", "adam_m", "adam_v"):
            t0, t1 = getattr(engines[0], name), getattr(engines[1], name)
            assert bool(torch.isfinite(t0).all()), (hid, name)
            assert torch.equal(t0, t1), (hid, name)
        assert engines[0].debug_get("handoff_failures") == 0.0
        assert all(np.isfinite(v) for v in engines[0].read_stats().values())


def _family_alg(family, B, seed):
    """(algorithm object, flat observation width, action dim) of the families the group surface serves besides DSAC_V2 over
    MLP nets: the configurations the reference itself runs sample_interval = 8 with are the CNN ones
    (example_train/dsacv2_cnn_carracing_offasync.py:133, dsacv1_cnn_carracing_offasync.py:133)."""
    if family == "v2_cnn":
        from test_hip_cnn_parity import make_pair as mk
        return mk((3, 96, 96), 3, "type_2", B, seed=seed)[0], 3 * 96 * 96, 3
    if family == "v1_mlp":
        from test_hip_v1_parity import make_pair as mk
        return mk(16, 4, (64, 64), B, seed=seed)[0], 16, 4
    if family == "v1_cnn":
        from test_hip_v1_cnn_parity import make_pair as mk
        return mk((3, 96, 96), 3, "type_2", B, seed=seed)[0], 3 * 96 * 96, 3
    raise KeyError(family)


@pytest.mark.parametrize("family,B,first,lengths", [
    ("v2_cnn", 16, 0, [8, 8, 3]),        # conv type_2 + twin-trunk chain units: groups replay the plain (29-launch) graph
    ("v2_cnn", 16, 3, [5, 8]),           # starts inside a delay_update period
    ("v1_mlp", 64, 1, [8, 3, 8]),        # DSAC_V1 on the row-slice chains: the pipelined graph
    ("v1_cnn", 16, 0, [8, 5]),
])
def test_run_group_equals_eager_steps_cnn_and_v1(family, B, first, lengths):
    """VERDICT r5 missing 2: dsact_run_group was pinned for DSAC_V2 over MLP nets only. The same bitwise statement -- group
    replays == { dsact_gather; dsact_step } per update: parameters, targets, both Adam moments, step state, statistics, the
    staged minibatch -- for the CNN approximators, DSAC_V1, and DSAC_V1 over the CNN approximators."""
    N = 64 if "cnn" in family else 2000
    engines = []
    for mode in ("eager", "group"):
        alg, O, A = _family_alg(family, B, seed=4)
        e = alg.engine
        e.set_device_rng(4242)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(9)
        e.buffer_fill_device(0, torch.rand(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) * 0.8 - 0.4,
                             torch.randn(N, device="cuda", generator=g), torch.rand(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .1).float())
        np.random.seed(7)
        it = first
        for n in lengths:
            rows = np.stack([np.random.randint(0, N, size=B) for _ in range(n)])
            if mode == "group":
                e.run_group(it, rows)
            else:
                for j in range(n):
                    e.gather(rows[j])
                    e.step(it + j)
            it += n
        e.sync()
        engines.append(e)
    same_engine_state(engines[0], engines[1], family + ": group vs eager")
    g = engines[1]
    assert torch.isfinite(g.online).all()
    if family == "v1_mlp":
        assert g.chain_active and g.debug_get("pipe_graph") == 1.0
    else:
        assert g.debug_get("pipe_graph") == 0.0          # CNN nets: pipe_eligible excludes them (csrc/dsact_api.hip)


def test_family_group_surface_equals_per_iteration_surface():
    """DSAC_V1_HIP inherits local_update_group: sample_batches + local_update_group == sample_batch + local_update per
    iteration through the plugin classes (NumPy stream, parameters, the last update's tb_info), strict RNG (the V1 noise
    draws travel as the noise table)."""
    from dsac_v1_hip import DSAC_V1_HIP
    from oracle.dsac_v1_oracle import V1_TB_KEYS
    from training.hip_replay_buffer import HipReplayBuffer

    O, A, hid, B, N, K = 16, 4, (64, 64), 64, 500, 8
    ring = host_ring(N, O, A, 9)
    out = []
    for mode in ("single", "group"):
        torch.manual_seed(2)
        kw = hip_kwargs(O, A, hid, B, buffer_max_size=N, seed=5, algorithm="DSAC_V1_HIP", TD_bound=10, strict_rng=True)
        alg = DSAC_V1_HIP(**kw)
        buf = HipReplayBuffer(**kw)
        assert buf.engine is alg.engine
        buf.add_batch([(ring["obs"][i], {}, ring["act"][i], float(ring["rew"][i]), ring["obs2"][i], bool(ring["done"][i]), 0.0, {})
                       for i in range(N)])
        np.random.seed(3)
        torch.manual_seed(77)
        if mode == "group":
            tb = alg.local_update_group(buf.sample_batches(B, K), 4)
            tb = alg.local_update_group(buf.sample_batches(B, 3), 4 + K)
        else:
            for it in range(4, 4 + K + 3):
                tb = alg.local_update(buf.sample_batch(B), it)
        out.append((alg, [float(tb[k]) for k in V1_TB_KEYS[:-1]], np.random.randint(0, 1 << 30), float(torch.rand(1))))
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]     # NumPy and torch generators stand where the loop leaves them
    assert out[0][1] == out[1][1]
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(out[0][0].engine, name), getattr(out[1][0].engine, name)), name


@pytest.mark.parametrize("first,n", [(1, 7), (0, 7), (1, 8), (3, 1), (1, 2)])
def test_fast_flag_groups_cut_anywhere(first, n):
    """with hip_flags = DSACT_F_SKIP_ACTOR_ON_OFF_ITERS a trainer's groups are cut at log / evaluation /
    checkpoint iterations, so they start and end anywhere relative to the delay_update period -- dsact_run_group refuses such a
    group. local_update_group now issues the misaligned head / tail one update at a time: == per-iteration local_update with
    the same flag, bit for bit."""
    from dsac_v2_hip import DSAC_V2_HIP
    from training.hip_replay_buffer import HipReplayBuffer

    O, A, hid, B, N = 16, 4, (64, 64), 64, 500
    ring = host_ring(N, O, A, 9)
    out = []
    for mode in ("single", "group"):
        torch.manual_seed(2)
        kw = hip_kwargs(O, A, hid, B, buffer_max_size=N, seed=5, hip_flags=1)
        alg = DSAC_V2_HIP(**kw)
        buf = HipReplayBuffer(**kw)
        buf.add_batch([(ring["obs"][i], {}, ring["act"][i], float(ring["rew"][i]), ring["obs2"][i], bool(ring["done"][i]), 0.0, {})
                       for i in range(N)])
        np.random.seed(3)
        if mode == "group":
            tb = alg.local_update_group(buf.sample_batches(B, n), first)
        else:
            for it in range(first, first + n):
                tb = alg.local_update(buf.sample_batch(B), it)
        out.append((alg, [float(tb[k]) for k in TB_KEYS[:-1]]))
    assert out[0][1] == out[1][1]
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(out[0][0].engine, name), getattr(out[1][0].engine, name)), name


def test_trainer_with_fast_flag_and_sample_interval_8(tmp_path):
    """the regression ADVICE r5 describes: sample_interval 8, delay_update 2, hip_flags 1 -- iteration 0 logs (a group of one), the
    next group of 7 starts at iteration 1. train() must run, and equal the ungrouped loop bit for bit."""
    import plugin

    finals = []
    for grouped in (True, False):
        kw = hip_kwargs(16, 4, (64, 64), 32, act_limit=0.3, env=_ToyEnv(), sample_batch_size=6, reward_scale=1,
                        buffer_warm_size=40, buffer_max_size=400, max_iteration=41, log_save_interval=12,
                        apprfunc_save_interval=1000, eval_interval=1000, ini_network_dir=None,
                        save_folder=str(tmp_path / ("g%d" % grouped)), seed=7, sample_interval=8, hip_flags=1,
                        hip_group_updates=grouped)
        torch.manual_seed(kw["seed"]); np.random.seed(kw["seed"])
        alg = plugin.create_alg(**kw)
        sampler = plugin.create_sampler(**kw)
        buf = plugin.create_buffer(**kw)
        tr = plugin.create_trainer(alg, sampler, buf, None, **kw)
        tr.train()
        alg.engine.sync()
        finals.append(alg.engine)
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(finals[0], name), getattr(finals[1], name)), name
    assert finals[0].get_state()["adam_steps"] == [41, 21, 21]
