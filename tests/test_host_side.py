"""CPU-side checks (no GPU): C-ABI exports, arena layout / cost model, plugin surface on CPU."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, hip_kwargs
from oracle.dsact_oracle import DsactOracle, default_config, policy_forward

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return g


def test_library_exports_every_declared_symbol(built):
    from dsact import _ffi
    lib = _ffi.load()
    hdr = open(os.path.join(ROOT, "include", "dsact.h")).read()
    declared = sorted(set(re.findall(r"\b(dsact_[a-z_0-9]+)\s*\(", hdr)))
    bound = sorted(n for n, _, _ in _ffi.SYMBOLS)
    assert declared == bound, set(declared) ^ set(bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dsact_version() >= 1
    # no GPU here: the library must say so instead of computing anything
    if lib.dsact_device_count() == 0:
        cfg = _ffi.Config()
        h = ctypes.c_void_p()
        assert lib.dsact_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -4  # DSACT_E_NODEVICE


def test_update_path_fails_loudly_without_gpu(built):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dsac_v2_hip import DSAC_V2_HIP
    from dsact._ffi import DsactError
    with pytest.raises(DsactError):
        DSAC_V2_HIP(**hip_kwargs(11, 3, (32, 32), 32))


def test_layout_counts_and_cost_model_match_survey():
    from dsact.layout import ArenaLayout
    l3 = ArenaLayout(376, 17, [256, 256, 256])
    assert (l3.n_q, l3.n_pi, l3.n_online - 1 + l3.n_target) == (232962, 236834, 1405516)
    f, b = l3.mac_per_sample()
    assert (f, b) == (1865216, 1375232)
    assert abs(l3.flop_per_step(256) / 1e9 - 1.659) < 1e-3
    assert abs(l3.bytes_per_step(256) / 1e6 - 23.25) < 0.02
    l2 = ArenaLayout(376, 17, [256, 256])
    f, b = l2.mac_per_sample()
    assert (f, b) == (1340928, 850944)
    assert abs(l2.bytes_per_step(256) / 1e6 - 16.93) < 0.02
    lay = json.load(open(os.path.join(GOLDEN, "checkpoint_layout.json")))
    assert [[k, list(v)] for k, v in l3.state_dict_keys().items()] == lay["humanoid_l3"]


def test_cpu_container_matches_reference_init_and_format():
    from dsac_v2_hip import ApproxContainer
    O, A, hid = 376, 17, (256, 256, 256)
    torch.manual_seed(0)
    net = ApproxContainer(**hip_kwargs(O, A, hid, 256))
    torch.manual_seed(0)
    orc = DsactOracle(default_config(O, A, hid))  # same init as the reference (tests/test_oracle_vs_reference.py)
    sd, osd = net.state_dict(), orc.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    for k in sd:
        assert torch.equal(sd[k], osd[k]), k
    obs = torch.randn(3, O)
    assert torch.equal(net.policy(obs), policy_forward(obs, [p.detach() for p in orc.p["policy"]], orc.cfg))
    dist = net.create_action_distributions(net.policy(obs))
    torch.manual_seed(5)
    a, lp = dist.sample()
    torch.manual_seed(5)
    x = torch.distributions.Normal(dist.mean, dist.std).sample()
    assert torch.equal(a, 0.4 * torch.tanh(x))
    assert a.shape == (3, A) and lp.shape == (3,)
    assert torch.allclose(dist.mode(), 0.4 * torch.tanh(dist.mean))
    lim = json.load(open(os.path.join(GOLDEN, "checkpoint_layout.json")))["pendulum_shipped"]
    pend = ApproxContainer(**hip_kwargs(3, 1, (256, 256, 256), 256, act_limit=2.0))
    assert [[k, list(v.shape)] for k, v in pend.state_dict().items()] == lim


def test_unsupported_configs_raise():
    from dsac_v2_hip import ApproxContainer
    with pytest.raises(NotImplementedError):
        ApproxContainer(**hip_kwargs(4, 2, (32,), 8, value_func_type="CNN"))
    with pytest.raises(NotImplementedError):
        ApproxContainer(**hip_kwargs(4, 2, (32,), 8, policy_hidden_activation="swish"))   # not one of the reference's six
    with pytest.raises(NotImplementedError):
        ApproxContainer(**hip_kwargs(4, 2, (32,), 8, value_output_activation="swish"))    # (every name of the reference's table is built)
    # output activations (networks/mlp.py:15-20: the module behind the last Linear) build the matching torch modules
    import torch
    from oracle.dsact_oracle import DsactOracle, default_config, policy_forward
    for oa in ("relu", "elu", "selu", "sigmoid", "tanh", "linear"):
        torch.manual_seed(3)
        c = ApproxContainer(**hip_kwargs(6, 2, (16, 16), 8, policy_output_activation=oa, value_output_activation=oa))
        orc = DsactOracle(default_config(6, 2, (16, 16), policy_out_act=oa, value_out_act=oa), state_dict=c.state_dict())
        x = torch.randn(5, 6)
        assert torch.equal(c.policy(x), policy_forward(x, orc.p["policy"], orc.cfg).detach()), oa
    # the reference's hidden activations (utils/common_utils.py:16-45) build the matching torch modules
    for act in ("relu", "elu", "selu", "sigmoid", "tanh", "gelu"):
        torch.manual_seed(3)
        c = ApproxContainer(**hip_kwargs(6, 2, (16, 16), 8, policy_hidden_activation=act, value_hidden_activation=act))
        orc = DsactOracle(default_config(6, 2, (16, 16), policy_act=act, value_act=act), state_dict=c.state_dict())
        x = torch.randn(5, 6)
        assert torch.equal(c.policy(x), policy_forward(x, orc.p["policy"], orc.cfg).detach()), act
    # another DEPTH than the policy's (round 6: each family keeps its own layer count)
    c = ApproxContainer(**hip_kwargs(4, 2, (32,), 8, value_hidden_sizes=[16, 16]))
    assert len(c.q1.q) == 6 and len(c.policy.policy) == 4 and c._layout.n_q == 16 * 6 + 16 + 16 * 16 + 16 + 2 * 16 + 2
    assert c._layout.n_pi == 32 * 4 + 32 + 4 * 32 + 4
    # value_hidden_sizes != policy_hidden_sizes of the same depth: the policy nets get their own widths
    c = ApproxContainer(**hip_kwargs(4, 2, (32, 32), 8, policy_hidden_sizes=[16, 24]))
    assert tuple(c.policy.policy[0].weight.shape) == (16, 4) and tuple(c.policy.policy[2].weight.shape) == (24, 16)
    assert tuple(c.q1.q[0].weight.shape) == (32, 6) and c._layout.n_pi == 16 * 4 + 16 + 24 * 16 + 24 + 4 * 24 + 4


def test_plugin_discovery_rules():
    import plugin
    assert plugin.camel("hip_replay_buffer") == "HipReplayBuffer"
    import importlib
    m = importlib.import_module("training.hip_replay_buffer")
    assert hasattr(m, "HipReplayBuffer")
    m = importlib.import_module("dsac_v2_hip")
    assert hasattr(m, "DSAC_V2_HIP") and hasattr(m, "ApproxContainer")


def test_cnn_arena_layout_views_are_disjoint_and_preserve_the_networks():
    """CNN nets: conv weights live in the arena as [Cout][KH][KW][Cin] and the twin
    mean/log_std output layers inside one (n_out x 2H) matrix; the state_dict tensors are strided views of it.
    Emulates ApproxContainer.attach on CPU arenas: no two parameters alias, values and forward passes survive."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "dsac-v2_amd"))
    import dsac_v2_hip as m
    from helpers import hip_kwargs

    obs_shape, A = (3, 96, 96), 3
    kw = hip_kwargs(obs_shape, A, (256, 256, 256), 8, act_limit=1.0)
    for key in ("value", "policy"):
        kw[key + "_func_type"] = "CNN"
        kw[key + "_conv_type"] = "type_2"
        kw.pop(key + "_hidden_sizes")
    torch.manual_seed(0)
    c = m.ApproxContainer(**kw)
    lay = c._layout
    assert lay.feat_dim == 256 and lay.n_online == 2 * lay.n_q + lay.n_pi + 1
    before = {k: v.clone() for k, v in c.state_dict().items()}
    obs = torch.rand(2, *obs_shape)
    act = torch.rand(2, A)
    with torch.no_grad():
        q_before, pi_before = c.q1(obs, act), c.policy(obs)
    arenas = {"online": torch.zeros(lay.n_online), "target": torch.zeros(lay.n_target)}
    owner = {"online": torch.full((lay.n_online,), -1, dtype=torch.long), "target": torch.full((lay.n_target,), -1, dtype=torch.long)}
    with torch.no_grad():
        for i, (p, arena, off, shape, strides) in enumerate(c._named_param_slots()):
            view = torch.as_strided(arenas[arena], shape, strides, off)
            ids = torch.as_strided(owner[arena], shape, strides, off)
            assert int((ids != -1).sum()) == 0, "parameter %d overlaps an earlier one" % i
            ids.fill_(i)
            view.copy_(p.data)
            p.data = view
    # everything not owned by a parameter is a structural zero of a twin output layer: 2*H floats per Q net,
    # 2*A*H per policy net (include/dsact.h)
    H = lay.hidden[-1]
    assert int((owner["online"] == -1).sum()) == 2 * (2 * H) + 2 * A * H
    assert int((owner["target"] == -1).sum()) == 2 * (2 * H) + 2 * A * H
    after = c.state_dict()
    assert list(after.keys()) == list(before.keys()) == list(lay.state_dict_keys().keys())
    for k in before:
        assert torch.equal(before[k], after[k]), k
    with torch.no_grad():
        # strided weights may take another BLAS path: values agree to rounding
        assert torch.allclose(c.q1(obs, act), q_before, atol=1e-6) and torch.allclose(c.policy(obs), pi_before, atol=1e-6)
    # conv weight memory order is [Cout][KH][KW][Cin]
    w = c.q1.conv[2].weight
    flat = arenas["online"][lay.param_views("q1")[2][2]:][: w.numel()].view(w.shape[0], w.shape[2], w.shape[3], w.shape[1])
    assert torch.equal(flat.permute(0, 3, 1, 2), w)


def test_v1_layout_matches_the_v1_oracle():
    from dsact.layout import ArenaLayout
    from oracle.dsac_v1_oracle import DsacV1Oracle

    lay = ArenaLayout(11, 3, [32, 32], n_critics=1)
    orc = DsacV1Oracle(default_config(11, 3, (32, 32)))
    assert lay.n_online == orc.flat_params().numel() and lay.n_target == orc.flat_targets().numel()
    assert list(lay.state_dict_keys().keys()) == list(orc.state_dict().keys())
    assert lay.online_nets == ("q", "policy")


def test_bench_helpers():
    """graph sizing replays exactly W and exactly K steps; roofline helper accounts every launch of the chain path"""
    import bench
    for steps, warm in ((20, 5), (20, 6), (4000, 400), (20000, 2000), (7, 3), (300, 0)):
        g = bench.graph_steps(steps, warm)
        assert 1 <= g <= 64 and steps % g == 0 and (warm % g == 0 or warm == 0), (steps, warm, g)
    assert bench.graph_steps(20, 6, even=True) % 2 == 0
    assert bench.n_regions(20) == 61 and bench.n_regions(4000) == 5 and bench.n_regions(20000) == 3
    from dsact.layout import ArenaLayout
    lay = ArenaLayout(376, 17, [256, 256, 256])
    fl = bench.chain_flops(lay, 256)
    # forward A+B = every forward MAC of SURVEY 8(d) (1,865,216 per sample); the shared obs part is counted once
    fwd = (fl["chain_fwd_a"] + fl["chain_fwd_b"]) / (2.0 * 256)
    assert abs(fwd - (1865216 - 2 * 376 * 256)) < 1, fwd
    assert bench.pmc_traffic("no-such-kernel") is None


def test_bench_gpus_flag_spawns_ranks_and_never_underreports(tmp_path):
    """`python bench.py --gpus 2` without a torch.distributed environment re-executes itself with 2 ranks (gloo dry run
    here: no GPU); without --dry-run-cpu and with fewer than N devices it refuses instead of printing n_gpus: 1."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["n_gpus"] == 2 and d["sum"] == 3.0 and d["steps"] == 20 and d["warmup"] == 5
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr, (r.returncode, r.stderr[-500:])
    assert "n_gpus" not in r.stdout


def test_hip_batch_detects_rows_overwritten_after_sampling():
    """a HipBatch token re-gathers by ring index when it is no longer the staged minibatch; the reference's
    batch is a copy taken at sample time (training/replay_buffer.py:85-90), so the token must notice that add_batch has
    replaced its rows (host-side logic only: a fake engine records the calls)."""
    from dsac_v2_hip import HipBatch

    class FakeEngine:
        def __init__(self, cap, ptr):
            self.buffer_capacity, self.buffer_ptr, self.rows_added, self.stage_serial, self.gathers = cap, ptr, 0, 0, 0
            self.fill_epoch = 0

        def gather(self, idx):
            self.gathers += 1
            self.stage_serial += 1

        def add(self, n):
            self.rows_added += n
            self.buffer_ptr = (self.buffer_ptr + n) % self.buffer_capacity

    e = FakeEngine(100, 90)
    e.stage_serial = 1
    tok = HipBatch(e, np.array([0, 5, 89, 95]))
    tok.restage()
    assert e.gathers == 0                       # it IS the staged minibatch
    e.stage_serial += 1                         # somebody else staged another one
    e.add(5)                                    # rows 90..94: none of the token's
    tok.restage()
    assert e.gathers == 1
    e.stage_serial += 1
    e.add(6)                                    # rows 95..99, 0: two of the token's rows are gone
    assert tok._overwritten() == 2
    with pytest.raises(RuntimeError, match="2 of the 4 sampled ring rows were overwritten"):
        tok.restage()
    e.add(200)
    assert tok._overwritten() == 4
    # buffer_fill_device writes at an arbitrary row (not in append order): every outstanding token is invalid
    e2 = FakeEngine(100, 10)
    tok2 = HipBatch(e2, np.array([50, 60]))
    assert tok2._overwritten() == 0
    e2.fill_epoch += 1
    assert tok2._overwritten() == 2


def test_one_vectorised_index_draw_is_the_stream_of_n_reference_draws():
    """HipReplayBuffer.sample_batches draws a group's rows with one np.random.randint(0, size, (n, batch)): values and the
    generator state afterwards must be those of n reference-style calls (training/replay_buffer.py:86), for ring sizes on both
    sides of 2**31 (the masked-rejection path changes width there)"""
    import numpy as np

    for size in (7, 1000, 10 ** 6, 10 ** 7, 2 ** 31 - 1, 2 ** 31 + 5, 2 ** 33):
        for n, b in ((8, 256), (3, 50), (1, 16)):
            np.random.seed(11)
            ref = np.stack([np.random.randint(0, size, size=b) for _ in range(n)])
            st_ref = np.random.get_state()
            np.random.seed(11)
            one = np.random.randint(0, size, size=(n, b))
            st_one = np.random.get_state()
            assert np.array_equal(ref, one) and ref.dtype == one.dtype
            assert np.array_equal(st_ref[1], st_one[1]) and st_ref[2:] == st_one[2:]
