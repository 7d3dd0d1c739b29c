"""The data-parallel coordinator (dsac-v2_amd/dsact/dp.py) over the gloo backend, world_size 2, on CPU.

The coordinator is product code; the engine plugged in here is an oracle-backed stand-in exposing the
same three members the HIP engine does (`grads`, `dp_grads()`, `dp_apply()`), because HIP kernels need
a GPU. Checks: (1) replicas stay bit-identical, (2) the averaged shard gradients equal the global-batch
gradients of a single process (to fp32 summation order), (3) parameters track the single-process oracle.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleDPEngine:
    """DsactOracle behind the DsactEngine data-parallel surface."""

    def __init__(self, orc, batches, noises):
        self.orc, self.batches, self.noises, self.k = orc, batches, noises, 0
        n = orc.flat_params().numel()
        self.grads = torch.zeros(n + 2)

    # strict mode: local std sums -> (coordinator all-reduces std_sums) -> loss with the global means
    def dp_set_strict(self, enable=True):
        self.std_sums = torch.zeros(2) if enable else None

    def dp_forward(self):
        b = self.batches[self.k]
        with torch.no_grad():
            _, s1 = self.orc._q(b["obs"], b["act"], self.orc.p["q1"])
            _, s2 = self.orc._q(b["obs"], b["act"], self.orc.p["q2"])
        self.std_sums[0], self.std_sums[1] = s1.sum(), s2.sum()

    def dp_backward(self):
        self.orc.std_mean_override = (self.std_sums[0] / self.global_batch, self.std_sums[1] / self.global_batch)
        self.dp_grads()
        self.orc.std_mean_override = None

    def dp_grads(self):
        self.orc.compute_gradient(self.batches[self.k], self.noises[self.k])
        self.grads[:-2] = self.orc.flat_grads()
        self.grads[-2] = float(self.orc.mean_std1)
        self.grads[-1] = float(self.orc.mean_std2)

    # overlapped mode: the critics' segment of the arena is final (and all-reduced) before the actor's
    @property
    def critic_grad_count(self):
        return sum(t.numel() for n in ("q1", "q2") for t in self.orc.p[n])

    def dp_grads_critic(self):
        self.orc.compute_gradient(self.batches[self.k], self.noises[self.k])
        n_c = self.critic_grad_count
        self.grads[:n_c] = self.orc.flat_grads()[:n_c]

    def dp_grads_actor(self):
        n_c = self.critic_grad_count
        self.grads[n_c:-2] = self.orc.flat_grads()[n_c:]
        self.grads[-2] = float(self.orc.mean_std1)
        self.grads[-1] = float(self.orc.mean_std2)

    def dp_apply(self):
        off = 0
        for n in ("q1", "q2", "policy"):
            for p in self.orc.p[n]:
                p.grad = self.grads[off:off + p.numel()].view_as(p).clone()
                off += p.numel()
        self.orc.log_alpha.grad = self.grads[off].clone()
        self.orc.mean_std1 = self.grads[-2].clone()
        self.orc.mean_std2 = self.grads[-1].clone()
        self.orc.update(self.k)
        self.k += 1


def _worker(rank, world, port, out_q, strict=False, overlap=False):
    for p in (ROOT, os.path.join(ROOT, "dsac-v2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from dsact.dp import DataParallelUpdater
    from helpers import synth_batch
    from oracle.dsact_oracle import DsactOracle, default_config, draw_noise

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    O, A, hid, B, steps = 11, 3, (32, 32), 32, 4
    cfg = default_config(O, A, hid)
    torch.manual_seed(100 + rank)  # different init per rank: the coordinator must broadcast rank 0's
    orc = DsactOracle(cfg)
    rng = np.random.default_rng(0)
    gb, gn = [], []
    for k in range(steps):
        gb.append(synth_batch(rng, B, O, A))
        torch.manual_seed(500 + k)
        gn.append(draw_noise(B, A))
    lo, hi = rank * B // world, (rank + 1) * B // world
    sb = [{k: v[lo:hi] for k, v in b.items()} for b in gb]
    sn = [{k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in n.items()} for n in gn]
    eng = OracleDPEngine(orc, sb, sn)
    eng.global_batch = B
    flat = [t for n in DsactOracle.NETS for t in orc.p[n]] + [orc.log_alpha]
    dp = DataParallelUpdater(eng, broadcast_tensors=[t.data for t in flat], strict=strict, overlap=overlap)
    assert dp.overlap == overlap
    grads0 = None
    for k in range(steps):
        if overlap:
            # the coordinator's own step(): critic half -> async all-reduce -> actor half -> all-reduce -> apply
            applied = eng.dp_apply
            eng.dp_apply = lambda: None
            dp.step()
            eng.dp_apply = applied
            if k == 0:
                grads0 = eng.grads.clone()
            eng.dp_apply()
            continue
        if strict:
            eng.dp_forward()
            dist.all_reduce(eng.std_sums, op=dist.ReduceOp.SUM)
            eng.dp_backward()
        else:
            eng.dp_grads()
        dp.allreduce_grads()
        if k == 0:
            grads0 = eng.grads.clone()
        eng.dp_apply()
    if rank == 0:
        # single-process reference at the global batch, same init (rank 0's)
        torch.manual_seed(100)
        ref = DsactOracle(cfg)
        ref.compute_gradient(gb[0], gn[0])
        g_ref = ref.flat_grads().clone()
        ref.update(0)
        for k in range(1, steps):
            ref.local_update(gb[k], gn[k], k)
        out_q.put(("ref", g_ref.numpy(), ref.flat_params().numpy(), float(ref.mean_std1)))
    out_q.put((rank, grads0.numpy(), orc.flat_params().numpy(), float(orc.mean_std1)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_global_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world + 1):
        item = q.get(timeout=180)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, p0, ms0 = got[0]
    g1, p1, ms1 = got[1]
    gr, pr, msr = got["ref"]
    # (1) replicas identical
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(g0, g1)
    assert ms0 == ms1
    # (2) first-step averaged gradient == global-batch gradient up to the local-vs-global mean_std used
    #     inside the loss on the sentinel step ("fast" mode) -- small but not rounding-level
    n = gr.size
    scale = np.abs(gr).max()
    assert np.abs(g0[:n] - gr).max() <= 2e-2 * scale
    # the re-synchronised EMA equals the global batch mean exactly (mean of equal-sized shard means)
    assert abs(ms0 - msr) <= 1e-6
    # (3) parameters after 4 updates track the single-process run (Adam moves ~lr per step)
    assert np.abs(p0 - pr).max() <= 5e-4


def test_two_rank_strict_mode_equals_global_batch_gradient():
    """strict=True: the 2-float pre-loss all-reduce makes the averaged shard gradients EQUAL the single-process
    global-batch gradients (fp32 summation order only), sentinel step included."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world + 1):
        item = q.get(timeout=180)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, p0, ms0 = got[0]
    g1, p1, ms1 = got[1]
    gr, pr, msr = got["ref"]
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(g0, g1)
    n = gr.size
    assert np.abs(g0[:n] - gr).max() <= 2e-6 * np.abs(gr).max() + 1e-9
    assert abs(ms0 - msr) <= 1e-6
    assert np.abs(p0 - pr).max() <= 2.1e-4   # Adam sign ambiguity of rounding-level gradients: 2*lr


def test_two_rank_overlapped_step_equals_plain_step():
    """overlap=True (critics' segment all-reduced asynchronously, the rest after the actor half) is the same
    arithmetic as the single all-reduce: replicas identical, same gradients and parameters as the plain run."""
    world = 2
    ctx = mp.get_context("spawn")
    runs = {}
    for overlap in (False, True):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, False, overlap)) for r in range(world)]
        for p in procs:
            p.start()
        got = {}
        for _ in range(world + 1):
            item = q.get(timeout=180)
            got[item[0]] = item[1:]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        runs[overlap] = got
    for r in (0, 1):
        for a, b in zip(runs[False][r], runs[True][r]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    np.testing.assert_array_equal(runs[True][0][1], runs[True][1][1])



def _fallback_worker(rank, world, port, out_q):
    for p in (ROOT, os.path.join(ROOT, "dsac-v2_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from dsact.dp import DataParallelUpdater

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Eng:     # the native surface the coordinator touches; the graph capture fails on rank 1 only
        def __init__(self):
            self.grads = torch.full((6,), float(rank + 1))
            self.calls = []

        def comm_unique_id(self):
            return b"\0" * 128

        def comm_init(self, r, w, uid):
            self.calls.append("comm_init")

        def dp_allreduce(self):       # stands in for the library's RCCL all-reduce (AVG)
            self.calls.append("allreduce")
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
            self.grads /= world

        def sync(self):
            pass

        def graph_build(self, n, flags):
            self.calls.append("graph_build")
            if rank == 1:
                raise RuntimeError("hipStreamEndCapture: operation not permitted when stream is capturing")

        def graph_run(self, it, n):
            self.calls.append("graph_run")

        def dp_begin(self, it):
            self.calls.append("dp_begin %d" % it)

        def dp_grads(self):
            self.calls.append("grads")

        def dp_apply(self):
            self.calls.append("apply")

    eng = Eng()
    dp = DataParallelUpdater(eng, native=True)
    ok = dp.build_graph(4)
    dp.run(8, 2)
    out_q.put((rank, ok, list(eng.calls), eng.grads.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_failed_graph_capture_falls_back_to_the_eager_coordinator_on_every_rank():
    """the first multi-GPU launch is also the first time ncclAllReduce with real peers is captured into a
    hipGraph. If the capture fails on ANY rank, EVERY rank must take the eager coordinator (same communicator) -- loudly --
    instead of one rank raising while the others hang in their first replayed collective."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        ok, calls, grads = got[r]
        assert ok is False
        assert calls[:3] == ["comm_init", "allreduce", "graph_build"]           # the communicator is warmed up before the capture
        assert "graph_run" not in calls
        assert calls[3:] == ["dp_begin 8", "grads", "allreduce", "apply", "grads", "allreduce", "apply"]
        assert grads == [1.5] * 6                                                # the eager collectives still average over both ranks
