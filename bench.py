#!/usr/bin/env python
"""bench.py -- DSAC-T gradient steps/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one DSAC_V2.local_update (reference dsac_v2.py:102-105) INCLUDING the replay gather
(training/replay_buffer.py:85-90): k_gather -> forward/backward of the six nets -> fused Adam/Polyak,
on synthetic Humanoid-shaped data (obs 376 / act 17, batch 256 per GPU, 1M-row replay ring resident in
HBM, random-init nets of the reference architecture). Nothing is skipped inside the timed region:
every step gathers a fresh minibatch, draws fresh noise (device Philox), computes all three losses and
their gradients, runs Adam for q1/q2 (policy/alpha/Polyak every `delay_update`-th step as the
reference does).

Prints ONE JSON line (rank 0). Extra objects: `roofline` (FP32-compute bound of the whole step, the
binding roofline per SURVEY.md section 8d), `roofline_hbm` (the HBM view north_star asks for),
`kernels` (per-launch hipEvent times of one eager step), `cpu_baseline` (the oracle port on the host
cores, bounded sample), `alt` (the other hidden-layer setting of SURVEY.md D1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dsac-v2_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md (vector == f32 MFMA)
HBM_PEAK_GBS = 8000.0     # spec; 6290 measured float4 copy

O, A, B = 376, 17, 256
N_REPLAY = 1_000_000
IDX_ROWS = 2048


def env_int(k, d):
    return int(os.environ.get(k, d))


def make_alg(hidden, device, seed=0, batch=B, v1=False):
    import numpy as np
    import torch
    from dsac_v2_hip import DSAC_V2_HIP

    torch.manual_seed(seed)
    kw = dict(
        algorithm="DSAC_V2_HIP", obsv_dim=O, action_dim=A, action_type="continu",
        value_func_type="MLP", policy_func_type="MLP", value_hidden_sizes=list(hidden),
        policy_hidden_sizes=list(hidden), value_hidden_activation="gelu", policy_hidden_activation="gelu",
        value_output_activation="linear", policy_output_activation="linear",
        policy_act_distribution="TanhGaussDistribution", policy_min_log_std=-20, policy_max_log_std=0.5,
        value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4,
        gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2, cnn_shared=False,
        replay_batch_size=batch, seed=seed + 1, hip_device=device,
        action_high_limit=np.full((A,), 0.4, np.float32), action_low_limit=np.full((A,), -0.4, np.float32),
    )
    if v1:
        from dsac_v1_hip import DSAC_V1_HIP

        return DSAC_V1_HIP(**dict(kw, algorithm="DSAC_V1_HIP", TD_bound=10))
    return DSAC_V2_HIP(**kw)


def fill_replay(engine, n_rows, seed):
    """synthetic ring per SURVEY.md 8(d): obs,obs2 ~ N(0,1); act ~ U(-.4,.4); rew ~ N(0,1); done ~ Bern(.01);
    generated on the device in chunks (3.09 GB at 1M rows never crosses PCIe)."""
    import torch

    engine.buffer_create(n_rows)
    g = torch.Generator(device=engine.device).manual_seed(seed)
    chunk = 131072
    for r0 in range(0, n_rows, chunk):
        n = min(chunk, n_rows - r0)
        obs = torch.randn(n, O, device=engine.device, generator=g)
        obs2 = torch.randn(n, O, device=engine.device, generator=g)
        act = torch.rand(n, A, device=engine.device, generator=g) * 0.8 - 0.4
        rew = torch.randn(n, device=engine.device, generator=g)
        done = (torch.rand(n, device=engine.device, generator=g) < 0.01).float()
        engine.buffer_fill_device(r0, obs, act, rew, obs2, done)


CNN_OBS, CNN_A, CNN_TYPE, CNN_ROWS = (3, 96, 96), 3, "type_2", 4096


def make_cnn_alg(device, seed=0, batch=B):
    """BASELINE.json configs[3]: DSAC_V2 with the reference's CNN approximators (networks/cnn.py, conv_type
    type_2 at the (3,96,96) CarRacing image the shipped example uses)."""
    import numpy as np
    import torch
    from dsac_v2_hip import DSAC_V2_HIP

    torch.manual_seed(seed)
    kw = dict(
        algorithm="DSAC_V2_HIP", obsv_dim=CNN_OBS, action_dim=CNN_A, action_type="continu",
        value_func_type="CNN", policy_func_type="CNN", value_conv_type=CNN_TYPE, policy_conv_type=CNN_TYPE,
        value_hidden_activation="gelu", policy_hidden_activation="gelu",
        value_output_activation="linear", policy_output_activation="linear",
        policy_act_distribution="TanhGaussDistribution", policy_min_log_std=-20, policy_max_log_std=0.5,
        value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4,
        gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2, cnn_shared=False,
        replay_batch_size=batch, seed=seed + 1, hip_device=device,
        action_high_limit=np.ones((CNN_A,), np.float32), action_low_limit=-np.ones((CNN_A,), np.float32),
    )
    return DSAC_V2_HIP(**kw)


def fill_replay_images(engine, n_rows, seed):
    """image ring: obs, obs2 ~ U[0,1) fp32 (pixel scale), act ~ U(-1,1), rew ~ N(0,1), done ~ Bern(.01)"""
    import torch

    engine.buffer_create(n_rows)
    g = torch.Generator(device=engine.device).manual_seed(seed)
    Oi, Ai = engine.obs_dim, engine.act_dim
    chunk = 1024
    for r0 in range(0, n_rows, chunk):
        n = min(chunk, n_rows - r0)
        obs = torch.rand(n, Oi, device=engine.device, generator=g)
        obs2 = torch.rand(n, Oi, device=engine.device, generator=g)
        act = torch.rand(n, Ai, device=engine.device, generator=g) * 2 - 1
        rew = torch.randn(n, device=engine.device, generator=g)
        done = (torch.rand(n, device=engine.device, generator=g) < 0.01).float()
        engine.buffer_fill_device(r0, obs, act, rew, obs2, done)


def cnn_cpu_baseline(batch, budget_s=8.0):
    import torch
    from oracle.dsact_oracle import draw_noise
    from oracle.dsact_oracle_cnn import DsactCnnOracle, cnn_config, synth_image_batch

    threads = 4
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    cfg = cnn_config(CNN_OBS, CNN_A, CNN_TYPE)
    orc = DsactCnnOracle(cfg)
    d = synth_image_batch(cfg, batch, seed=0)
    orc.local_update(d, draw_noise(batch, CNN_A), 0)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s or steps < 2:
        orc.local_update(d, draw_noise(batch, CNN_A), 1 + steps)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d updates (batch %d, fixed minibatch, no replay gather) in %.1f s, torch %s CPU, %d of %d host cores"
                      % (steps, batch, dt, torch.__version__, threads, os.cpu_count())}


def bench_cnn(device, steps, warmup, batch=B, cpu=True):
    """configs[3] as a secondary measurement (the headline metric is the Humanoid MLP workload)."""
    alg = make_cnn_alg(device, seed=0, batch=batch)
    e = alg.engine
    fill_replay_images(e, CNN_ROWS, seed=100)
    upload_indices(e, CNN_ROWS, 256, seed=1)
    wall, ev_ms = measure(alg, steps, warmup)
    stats = e.read_stats()
    lay = e.layout
    flop, byts = lay.flop_per_step(batch), lay.bytes_per_step(batch, 2)
    sps = steps / wall
    out = {
        "workload": "gym_carracingraw-shaped DSAC_V2 update: image %s fp32, act %d, conv %s + twin 256x3 MLPs, batch %d, "
                    "%d-row image replay ring in HBM" % ("x".join(map(str, CNN_OBS)), CNN_A, CNN_TYPE, batch, CNN_ROWS),
        "value": sps, "unit": "steps/s", "ms_per_step": 1000.0 * wall / steps, "steps": steps,
        "finite_stats": all(v == v and abs(v) < 1e30 for v in stats.values()),
        "roofline_step": {"bound": "mfma", "achieved": flop * sps / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": flop * sps / 1e12 / FP32_PEAK_TFLOPS, "flop_per_step": flop},
        "roofline_hbm": {"bound": "hbm", "achieved": byts * sps / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byts * sps / 1e9 / HBM_PEAK_GBS, "bytes_per_step": byts,
                         "note": "algorithmic: image gather 2*B*C*H*W*4 + weights + Adam/Polyak streams; activations assumed on-chip"},
    }
    try:
        prof = e.profile_step(warmup + steps)
        e.sync()
        out["kernels"] = [{"name": n, "us": round(ms * 1000, 2), "blocks": b} for n, ms, b in prof]
    except Exception as ex:
        out["kernels_error"] = str(ex)
    if cpu:
        out["cpu_baseline"] = cnn_cpu_baseline(batch)
    return out


def pmc_traffic_forward_stage():
    """HBM-side bytes per launch of the forward tile stage from the committed PMC passes (profiles/r01_pmc_traffic.json,
    produced by scripts/gpu_pmc2.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench, FETCH_SIZE doubled per the
    MI355X guide's gfx950 note). Not measurable live (needs rocprofv3); None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        d = json.load(open(path))["mlp"]
    except (OSError, KeyError, ValueError):
        return None
    tot, n = 0.0, 0
    for k, v in d.items():
        if "k_stage<false, false, 0" in k:
            tot += (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 * v["launches"]
            n += v["launches"]
    return tot / n if n else None


def upload_indices(engine, n_rows, rows, seed):
    import numpy as np

    np.random.seed(seed)  # legacy global RandomState, as the reference buffer uses (replay_buffer.py:86)
    engine.upload_index_table(np.random.randint(0, n_rows, size=(rows, engine.batch)))


def cpu_baseline(hidden, budget_s=12.0):
    """The oracle port (oracle/dsact_oracle.py == reference arithmetic, pinned bit-exact against the live
    reference in tests/test_oracle_vs_reference.py) timed on the host cores: sample_batch + local_update,
    4 torch threads like the reference (utils/init_args.py:14)."""
    import numpy as np
    import torch
    from oracle.dsact_oracle import DsactOracle, ReplayOracle, default_config, draw_noise

    threads = 4
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    orc = DsactOracle(default_config(O, A, hidden))
    n = 100_000
    buf = ReplayOracle(O, A, n)
    rng = np.random.default_rng(0)
    buf.buf["obs"][:] = rng.standard_normal((n, O), dtype=np.float32)
    buf.buf["obs2"][:] = rng.standard_normal((n, O), dtype=np.float32)
    buf.buf["act"][:] = rng.uniform(-0.4, 0.4, (n, A)).astype(np.float32)
    buf.buf["rew"][:] = rng.standard_normal(n, dtype=np.float32)
    buf.buf["done"][:] = (rng.random(n) < 0.01).astype(np.float32)
    buf.size = n
    np.random.seed(1)
    it = 0
    for _ in range(10):
        orc.local_update(buf.sample_batch(B), draw_noise(B, A), it)
        it += 1
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        orc.local_update(buf.sample_batch(B), draw_noise(B, A), it)
        it += 1
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d updates (batch 256, hidden %s, 100k-row host ring) in %.1f s, torch %s CPU, %d of %d host cores"
                      % (steps, "x".join(map(str, hidden)), dt, torch.__version__, threads, os.cpu_count())}


def boundary_rates(alg, n):
    """The drop-in boundary when the minibatch is NOT in the HIP ring: `local_update(data, it)` with the reference
    ReplayBuffer's CPU tensors (PCIe-inclusive: pageable host -> HBM every step) and with the `.cuda()` tensors the
    reference trainer makes of them (training/trainer.py:72-74; device-to-device). Eager launches, host-synchronous
    staging -- reported beside the headline, never as `value`."""
    import numpy as np
    import torch

    e = alg.engine
    rng = np.random.default_rng(5)
    O, A, Bt = e.obs_dim, e.act_dim, e.batch
    cpu = {"obs": torch.as_tensor(rng.standard_normal((Bt, O), dtype=np.float32)),
           "obs2": torch.as_tensor(rng.standard_normal((Bt, O), dtype=np.float32)),
           "act": torch.as_tensor(rng.uniform(-0.4, 0.4, (Bt, A)).astype(np.float32)),
           "rew": torch.as_tensor(rng.standard_normal(Bt, dtype=np.float32)),
           "done": torch.zeros(Bt)}
    res = {}
    for name, d in (("host_batch", cpu), ("cuda_batch", {k: v.cuda(e.device_index) for k, v in cpu.items()})):
        for it in range(20):
            alg.local_update(d, it)
        e.sync()
        t0 = time.perf_counter()
        for it in range(n):
            alg.local_update(d, it)
        e.sync()
        w = time.perf_counter() - t0
        res[name] = {"value": n / w, "unit": "steps/s", "ms_per_step": 1000.0 * w / n}
    res["bytes_per_step"] = 4 * Bt * (2 * O + A + 2)
    res["note"] = ("local_update(data) with a minibatch from outside the HIP ring, %d eager updates each: host_batch = "
                   "CPU tensors (PCIe-inclusive), cuda_batch = CUDA tensors; not the headline value" % n)
    return res


def graph_steps(steps, warmup, cap=64):
    """updates captured per hipGraph: the largest even divisor (<= cap) of both the timed and the warm-up step count.
    Measured: 2 -> 9,248, 8 -> 9,416, 40 -> 9,473 steps/s (the gap between graph launches amortises)."""
    import math

    if os.environ.get("DSACT_BENCH_GRAPH_STEPS"):
        return int(os.environ["DSACT_BENCH_GRAPH_STEPS"])
    g = math.gcd(int(steps), int(warmup)) if warmup else int(steps)
    best = 2
    for d in range(2, cap + 1, 2):
        if g % d == 0:
            best = d
    return best


def measure(alg, steps, warmup, world=1, dp=None, flags=0):
    """returns (wall seconds for `steps` steps, hipEvent ms for the same region or None)"""
    import torch

    e = alg.engine
    if dp is None:
        e.graph_build(graph_steps(steps, warmup), flags)
        e.graph_run(0, warmup)
        e.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms = e.time_steps(warmup, steps, use_graph=True)  # hipEvents on the engine's stream + host sync
        wall = time.perf_counter() - t0
        return wall, ms
    import torch.distributed as dist

    e.dp_begin(0)
    for _ in range(warmup):
        dp.step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dp.step()
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    return wall, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--hidden", type=str, default="256,256,256",
                    help="reference default (example_train/*.py); BASELINE.json words it as 256,256 -> reported in `alt`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true")
    ap.add_argument("--fast", action="store_true", help="primary number with DSACT_F_SKIP_ACTOR_ON_OFF_ITERS (default: strict; fast is reported in `fast`)")
    ap.add_argument("--replay-rows", type=int, default=N_REPLAY)
    ap.add_argument("--batch", type=int, default=B, help="minibatch rows per GPU (BASELINE metric: 256)")
    ap.add_argument("--cnn-only", action="store_true", help="measure only the CNN workload (configs[3]); prints its object")
    ap.add_argument("--cnn-steps", type=int, default=400)
    ap.add_argument("--cnn-type", type=str, default="type_2", help="type_2 at (3,96,96) (default) or type_1 at (4,84,84) (SURVEY.md D3)")
    args = ap.parse_args()
    steps = args.steps + (args.steps & 1)
    warmup = args.warmup + (args.warmup & 1)

    import __graft_entry__ as entry
    rank, world, local = env_int("RANK", 0), env_int("WORLD_SIZE", 1), env_int("LOCAL_RANK", 0)
    if rank == 0:
        entry.build()
    import torch

    dp = None
    # DSACT_BENCH_FORCE_DP=1: take the data-parallel (RCCL) code path even with one rank -- the only way to
    # exercise it on a 1-GPU box
    use_dp = world > 1 or os.environ.get("DSACT_BENCH_FORCE_DP") == "1"
    if use_dp:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist.barrier()
        if rank != 0:
            entry.build()
    if args.cnn_only:
        if args.cnn_type == "type_1":
            global CNN_OBS, CNN_A, CNN_TYPE
            CNN_OBS, CNN_A, CNN_TYPE = (4, 84, 84), 3, "type_1"
        print(json.dumps({"cnn": bench_cnn(local, args.cnn_steps, 40, cpu=not args.no_cpu_baseline)}))
        return
    hidden = [int(x) for x in args.hidden.split(",")]
    alg = make_alg(hidden, local, seed=0, batch=args.batch)
    e = alg.engine
    fill_replay(e, args.replay_rows, seed=100 + rank)  # every rank owns its own replay shard
    upload_indices(e, args.replay_rows, IDX_ROWS, seed=1 + rank)
    if use_dp:
        from dsact.dp import DataParallelUpdater

        # the updater issues its collectives on the engine's own stream (engine.torch_stream)
        # default: ONE all-reduce after the whole backward. DSACT_DP_OVERLAP=1 all-reduces the critics' 2/3 of the
        # arena asynchronously under the actor's backward -- measured on one rank the second collective call and
        # its cross-stream events cost +40 us/step against +14.5 us for the single call, so it is opt-in
        dp = DataParallelUpdater(e, broadcast_tensors=(e.online, e.target, e.adam_m, e.adam_v),
                                 overlap=os.environ.get("DSACT_DP_OVERLAP", "0") == "1")
        dp.force_collective = os.environ.get("DSACT_DP_FORCE_COLLECTIVE") == "1"
    wall, ev_ms = measure(alg, steps, warmup, world, dp, flags=1 if args.fast else 0)
    if use_dp:
        import torch.distributed as dist

        t = torch.tensor([wall], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    stats = e.read_stats()
    finite = all(v == v and abs(v) < 1e30 for v in stats.values())
    updates_per_s = steps / wall
    value = updates_per_s * world  # batch-256 gradient-step equivalents per second over the whole job
    lay = e.layout
    Bb = args.batch
    flop = lay.flop_per_step(Bb)
    byts = lay.bytes_per_step(Bb, 2)
    out = {
        "metric": "DSAC-T gradient steps/sec, batch=%d Humanoid (obs376/act17)" % args.batch,
        "value": value, "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1000.0 * wall / steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "gym_humanoid-shaped DSAC_V2 update: obs 376, act 17, MLP %s GELU, batch %d per GPU, "
                        "%d-row replay ring in HBM per GPU, gather+forward+backward+Adam+Polyak every step"
                        % ("x".join(map(str, hidden)), args.batch, args.replay_rows),
            "global_batch": args.batch * world, "parallelism": "dp%d" % world, "hidden": hidden,
            "noise": "device Philox4x32-10", "mode": "fast (discarded actor backward skipped)" if args.fast else "strict (every gradient the reference computes)",
            "launch": ("hipGraph (%d steps/graph; the next update's gather rides in the loss launch)" % graph_steps(steps, warmup)) if not use_dp else ("eager + RCCL all-reduce (%s)" % ("critics' segment overlapped with the actor backward" if dp.overlap else "single")),
            "unit_note": "value = synchronized updates/s x n_gpus (each rank contributes one batch-256 gradient per update)",
        },
        "finite_stats": finite,
    }
    if rank == 0:
        per_gpu_steps = updates_per_s
        # dominant kernel = k_stage<KC,KC,bias+GELU> (the forward tile stages: 6 of the 15 launches, ~37% of the
        # update): every forward stage is launched back to back on the engine's stream between two hipEvents
        n_st = 2 * len(hidden)
        st_ms, st_macs, reps = 0.0, 0.0, 300
        for st in range(n_st):
            ms_i, macs_i = e.time_stage(st, reps)
            st_ms += ms_i
            st_macs += macs_i
        dur_us = 1000.0 * st_ms / (reps * n_st)
        flop_launch = 2.0 * st_macs / n_st
        out["roofline"] = {
            "bound": "mfma", "achieved": flop_launch / (dur_us * 1e-6) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_launch / (dur_us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, "traffic": pmc_traffic_forward_stage(),
            "kernel": "dsact::k_stage<false,false,0,{0,4}> (forward tile stages: bias+GELU epilogue, 4 GEMM problems per launch)",
            "avg_launch_us": dur_us, "flop_per_launch": flop_launch,
            "note": "fp32 MFMA peak; avg over the %d forward stages, %d back-to-back launches each (hipEvents on the engine's "
                    "stream); algorithmic FLOP = 2*M*N*K of the stage's problems. traffic = bytes/launch (2*FETCH_SIZE + WRITE_SIZE) "
                    "from the committed PMC passes (profiles/r01_pmc_summary.txt): ~5.4 MB against 4.2 MB of operands + outputs "
                    "-- memory-side traffic is not what bounds this kernel" % (n_st, reps),
        }
        out["roofline_step"] = {
            "bound": "mfma", "achieved": flop * per_gpu_steps / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop * per_gpu_steps / 1e12 / FP32_PEAK_TFLOPS,
            "note": "whole update (one launch chain); algorithmic FLOP/step = %.4g (SURVEY.md 8d)" % flop,
        }
        out["roofline_hbm"] = {
            "bound": "hbm", "achieved": byts * per_gpu_steps / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": byts * per_gpu_steps / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "note": "algorithmic bytes/step = %.4g (SURVEY.md 8d)" % byts,
        }
        if ev_ms is not None:
            out["hip_event_ms_per_step"] = ev_ms / steps
        try:
            prof = e.profile_step(warmup + steps)
            e.sync()
            out["kernels"] = [{"name": n, "us": round(ms * 1000, 2), "blocks": b} for n, ms, b in prof]
            out["kernels_note"] = ("hipEvent-bracketed launches of ONE EAGER update (own gather launch, event overhead "
                                   "included); the timed graph replay has no per-update gather (it rides in `loss`) -- "
                                   "in-graph durations: profiles/r01_final_step_trace.txt")
            dom = max(prof, key=lambda r: r[1])
            out["dominant_kernel"] = {"name": dom[0], "us": round(dom[1] * 1000, 2)}
        except Exception as ex:  # profiling is informational
            out["kernels_error"] = str(ex)
    if rank == 0 and not use_dp and not args.fast:
        # same workload with the actor/alpha backward skipped on the off iterations of the delayed update: the
        # reference computes and discards those gradients (dsac_v2.py:174-186 vs :324); bitwise-identical parameter
        # trajectory (tests/test_hip_parity.py::test_skip_discarded_actor_backward_keeps_trajectory)
        wf, _ = measure(alg, steps, warmup, flags=1)
        out["fast"] = {"value": steps / wf, "unit": "steps/s", "ms_per_step": 1000.0 * wf / steps,
                       "note": "DSACT_F_SKIP_ACTOR_ON_OFF_ITERS; not the headline value"}
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        try:
            out["boundary"] = boundary_rates(alg, min(steps, 600))
        except Exception as ex:  # informational leg
            out["boundary_error"] = repr(ex)
    if not use_dp and not args.no_alt and args.batch == B:
        alt_hidden = [256, 256] if hidden != [256, 256] else [256, 256, 256]
        del alg
        alg2 = make_alg(alt_hidden, local, seed=0)
        fill_replay(alg2.engine, min(args.replay_rows, 200_000), seed=100)
        upload_indices(alg2.engine, min(args.replay_rows, 200_000), IDX_ROWS, seed=1)
        w2, _ = measure(alg2, steps, warmup)
        l2 = alg2.engine.layout
        out["alt"] = {"hidden": alt_hidden, "value": steps / w2, "unit": "steps/s",
                      "frac_fp32": l2.flop_per_step(B) * steps / w2 / 1e12 / FP32_PEAK_TFLOPS,
                      "frac_hbm": l2.bytes_per_step(B, 2) * steps / w2 / 1e9 / HBM_PEAK_GBS}
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        # the reference's DSAC_V1 (one critic) on the same kernels, same shapes
        try:
            alg1 = make_alg(hidden, local, seed=0, v1=True)
            fill_replay(alg1.engine, min(args.replay_rows, 200_000), seed=100)
            upload_indices(alg1.engine, min(args.replay_rows, 200_000), IDX_ROWS, seed=1)
            w1, _ = measure(alg1, steps, warmup)
            l1 = alg1.engine.layout
            out["dsac_v1"] = {"value": steps / w1, "unit": "steps/s", "ms_per_step": 1000.0 * w1 / steps,
                              "frac_fp32": l1.flop_per_step(B) * steps / w1 / 1e12 / FP32_PEAK_TFLOPS}
            del alg1
        except Exception as ex:
            out["dsac_v1_error"] = repr(ex)
    if rank == 0 and not use_dp and not args.no_cpu_baseline and args.batch == B:
        out["cpu_baseline"] = cpu_baseline(hidden)
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        try:
            out["cnn"] = bench_cnn(local, args.cnn_steps, 40, cpu=not args.no_cpu_baseline)
        except Exception as ex:  # secondary workload: never costs the headline line
            out["cnn_error"] = repr(ex)
    if rank == 0:
        print(json.dumps(out))
    if use_dp:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
