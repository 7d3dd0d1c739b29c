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

`--gpus N` without a torch.distributed environment re-executes itself under torch.distributed.run with N ranks
(one per GPU) and fails loudly when fewer than N devices exist. N > 1: every rank owns its replay shard and its
sampler-side state; the data-parallel update (gather -> gradients -> RCCL all-reduce -> Adam/Polyak) is captured
in ONE hipGraph per rank through the library's own communicator (dsact_comm_init), BASELINE.json configs[4].

Timing: W untimed warm-up steps, then R timed regions of EXACTLY K steps each, every region bracketed by a
barrier + device synchronisation on both sides, MAX over ranks per region; `value` is the MEDIAN region
(R = 61 for K <= 100, 5 for K <= 10000, else 3; all regions are listed in `regions_ms`).

Prints ONE JSON line (rank 0). Extra objects: `roofline` (the dominant kernel: algorithmic FLOP per launch / its
average in-chain duration, measured live with the dispatch's own start/stop events), `roofline_step` (whole
update vs the FP32 peak, the binding roofline per SURVEY.md section 8d), `roofline_hbm` (the HBM view north_star
asks for), `kernels`, `cpu_baseline` (the unmodified reference when it is mounted, else the oracle port; 1M-row
host buffer, 4 threads and all cores), `alt` (the other hidden-layer setting of SURVEY.md D1).
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


def bench_shapes(device, steps=4000):
    """Other shapes of the same update on the row-slice chains since round 6 (informational; NOT the headline workload): MuJoCo
    observation widths that are no multiple of 4, ragged / unequal hidden widths stored zero-padded, each beside
    the form rounds 1-5 ran it in (tile-stage kernels). Graph replays on a 100k-row synthetic ring, batch 256."""
    import numpy as np
    import torch
    from dsac_v2_hip import DSAC_V2_HIP

    cases = [("halfcheetah_walker2d obs17 act6 3x256", 17, 6, [256] * 3, None),
             ("hopper obs11 act3 3x256", 11, 3, [256] * 3, None),
             ("humanoid critics 3x256 policy 3x128", O, A, [256] * 3, [128] * 3),
             ("humanoid 3x200", O, A, [200] * 3, None)]
    out = []
    N = 100_000
    for name, o, a, hv, hp in cases:
        row = {"shape": name}
        for form in ("chains", "tile_stages"):
            if form == "tile_stages":
                os.environ["DSACT_NO_CHAIN"] = "1"
            try:
                torch.manual_seed(0)
                alg = DSAC_V2_HIP(
                    algorithm="DSAC_V2_HIP", obsv_dim=o, action_dim=a, action_type="continu", value_func_type="MLP", policy_func_type="MLP",
                    value_hidden_sizes=list(hv), policy_hidden_sizes=list(hp or hv), value_hidden_activation="gelu",
                    policy_hidden_activation="gelu", value_output_activation="linear", policy_output_activation="linear",
                    policy_act_distribution="TanhGaussDistribution", policy_min_log_std=-20, policy_max_log_std=0.5,
                    value_learning_rate=1e-4, policy_learning_rate=1e-4, alpha_learning_rate=3e-4, gamma=0.99, tau=0.005, auto_alpha=True,
                    alpha=0.2, delay_update=2, cnn_shared=False, replay_batch_size=B, seed=1, hip_device=device,
                    hip_pad_widths=form == "chains",
                    action_high_limit=np.full((a,), 0.4, np.float32), action_low_limit=np.full((a,), -0.4, np.float32))
            finally:
                os.environ.pop("DSACT_NO_CHAIN", None)
            e = alg.engine
            e.set_device_rng(1)
            e.buffer_create(N)
            g = torch.Generator(device=e.device).manual_seed(1)
            e.buffer_fill_device(0, torch.randn(N, o, device=e.device, generator=g), torch.rand(N, a, device=e.device, generator=g) * 0.8 - 0.4,
                                 torch.randn(N, device=e.device, generator=g), torch.randn(N, o, device=e.device, generator=g),
                                 (torch.rand(N, device=e.device, generator=g) < 0.01).float())
            np.random.seed(1)
            e.upload_index_table(np.random.randint(0, N, size=(64, B)))
            e.graph_build(8)
            e.time_steps(0, 400, use_graph=True)
            ms = min(e.time_steps(400 + steps * k, steps, use_graph=True) for k in range(2))
            assert e.chain_active == (form == "chains")
            row[form] = {"value": 1000.0 * steps / ms, "unit": "steps/s", "us_per_step": 1000.0 * ms / steps}
            if form == "chains":
                row["stored_width"] = e.layout.pad_to or hv[0]
            e.sync()
            e.close()
        out.append(row)
    return out


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
    wall, _, ev_ms = measure(alg, steps, warmup)
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
        out["launches_per_update"] = len(prof)
    except Exception as ex:
        out["kernels_error"] = str(ex)
    # memory-side bytes per update from the committed PMC passes of this workload (2 * FETCH_SIZE + WRITE_SIZE per launch x launches
    # per update, summed over kernels; scripts/gpu_final.sh) -- NOT measured in this run
    try:
        path = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r05_pmc_traffic_cnn.json", "r04_pmc_traffic_cnn.json")) if os.path.exists(q)), "")
        per = json.load(open(path))["cnn"]
        n_upd = max(v["launches"] for k, v in per.items() if "k_gather_img" in k)
        tot = sum((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 * v["launches"] / n_upd for v in per.values())
        out["roofline_hbm"]["traffic"] = tot
        out["roofline_hbm"]["traffic_source"] = "profiles/%s (committed rocprofv3 --pmc passes at batch 256; not measured in this run)" % os.path.basename(path)
    except Exception:
        out["roofline_hbm"]["traffic"] = None
    if cpu:
        out["cpu_baseline"] = cnn_cpu_baseline(batch)
    return out


def upload_indices(engine, n_rows, rows, seed):
    import numpy as np

    np.random.seed(seed)  # legacy global RandomState, as the reference buffer uses (replay_buffer.py:86)
    engine.upload_index_table(np.random.randint(0, n_rows, size=(rows, engine.batch)))


CPU_ROWS = int(os.environ.get("DSACT_CPU_ROWS", 100_000))   # host buffer of the CPU leg: first touch of fresh pages costs ~20 MB/s in this sandbox (a 1M-row,
                      # 3.1 GB buffer = minutes); sample_batch is 2 % of the CPU step even at 1M rows


def _cpu_fill(bufs, n_rows):
    """SURVEY.md 8(d) buffer recipe written IN PLACE into the buffer's own arrays: np.random.default_rng(0); obs, obs2 ~
    N(0,1); act ~ U(-.4,.4); rew ~ N(0,1); done ~ Bernoulli(.01); logp = 0"""
    import numpy as np

    rng = np.random.default_rng(0)
    for k in ("obs", "obs2"):
        rng.standard_normal(out=bufs[k].reshape(-1), dtype=np.float32)
    bufs["act"][:] = rng.uniform(-0.4, 0.4, (n_rows, A)).astype(np.float32)
    rng.standard_normal(out=bufs["rew"].reshape(-1), dtype=np.float32)
    bufs["done"][:] = (rng.random(n_rows) < 0.01).astype(np.float32).reshape(bufs["done"].shape)


def cpu_baseline_child(hidden, n_rows=CPU_ROWS, warm=50, timed=300, budget_s=12.0):
    """The reference CPU path beside the GPU number (BASELINE.md section 3 / SURVEY.md 8d): the UNMODIFIED reference
    (`create_alg` -> DSAC_V2, `create_buffer` -> ReplayBuffer, imported in place through oracle/ref_loader.py) when
    /root/reference is mounted -- kind "reference"; otherwise (the GPU box) the oracle port of the same arithmetic,
    pinned bit-exact against it in tests/test_oracle_vs_reference.py -- kind "port". Timed loop =
    sample_batch(B) + local_update, `warm` warm-up + up to `timed` updates, every loop bounded by budget_s, at 4 torch
    threads (the reference's own setting, utils/init_args.py:14) and at os.cpu_count() threads (skipped when one
    update at that thread count takes longer than 2 s: an over-subscribed container)."""
    import numpy as np
    import torch
    from oracle import ref_loader

    use_ref = ref_loader.reference_available()
    if use_ref:
        ref_loader.import_reference()
        from utils.initialization import create_alg, create_buffer   # the reference's own factories

        kw = ref_loader.reference_kwargs(O, A, tuple(hidden), buffer_max_size=n_rows, replay_batch_size=B)
        torch.manual_seed(0)
        alg = create_alg(**kw)
        buf = create_buffer(**kw)
        _cpu_fill(buf.buf, n_rows)
        buf.size, buf.ptr = n_rows, 0

        def step(it):
            alg.local_update(buf.sample_batch(B), it)
    else:
        from oracle.dsact_oracle import DsactOracle, ReplayOracle, default_config, draw_noise

        torch.manual_seed(0)
        orc = DsactOracle(default_config(O, A, hidden))
        buf = ReplayOracle(O, A, n_rows)
        _cpu_fill(buf.buf, n_rows)
        buf.size = n_rows

        def step(it):
            orc.local_update(buf.sample_batch(B), draw_noise(B, A), it)
    legs = {}
    ncpu = os.cpu_count() or 4
    it = 0
    def probe(threads):   # a cheap stand-in for one layer: is this thread count usable at all in this container?
        torch.set_num_threads(threads)
        x, w = torch.randn(256, 512), torch.randn(512, 256)
        (x @ w).sum().item()
        t0 = time.perf_counter()
        for _ in range(20):
            torch.nn.functional.gelu(x @ w)
        return time.perf_counter() - t0

    p4 = probe(4)
    for threads in sorted({4, ncpu}):
        if threads != 4:
            pn = probe(threads)
            if pn > 3.0 * p4:
                legs[threads] = {"skipped": "%d torch threads are over-subscribed in this container: a 256x512x256 layer runs %.0fx slower "
                                            "than at 4 threads" % (threads, pn / p4)}
                continue
        torch.set_num_threads(threads)
        np.random.seed(1)
        step(it)   # thread-pool start-up
        it += 1
        t0 = time.perf_counter()
        step(it)
        it += 1
        one = time.perf_counter() - t0
        if one > 2.0:
            legs[threads] = {"skipped": "one update takes %.1f s at %d torch threads" % (one, threads)}
            continue
        t0 = time.perf_counter()
        nw = 2
        while nw < warm and time.perf_counter() - t0 < budget_s / 2:
            step(it)
            it += 1
            nw += 1
        # three timed repeats (the hosts are shared: the spread is part of the answer), each a third of the budget
        reps, n_all, dt_all = [], 0, 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            n = 0
            while n < max(timed // 3, 1) and time.perf_counter() - t0 < budget_s / 3.0:
                step(it)
                it += 1
                n += 1
            dt = time.perf_counter() - t0
            reps.append(n / dt)
            n_all += n
            dt_all += dt
        legs[threads] = {"value": sorted(reps)[1], "updates": n_all, "seconds": dt_all, "warmup": nw,
                         "repeats": [round(v, 2) for v in reps], "min": min(reps), "max": max(reps)}
    v4 = legs[4]
    who = ("unmodified reference (create_alg -> DSAC_V2, create_buffer -> ReplayBuffer)" if use_ref else
           "oracle port of the reference arithmetic (the reference is not mounted on this box)")
    try:
        e2e = e2e_cpu(hidden)
    except Exception as ex:   # informational leg
        e2e = {"error": repr(ex)}
    return {"value": v4.get("value"), "unit": "steps/s", "cores": 4, "kind": "reference" if use_ref else "port",
            "repeats": v4.get("repeats"), "min": v4.get("min"), "max": v4.get("max"),
            "all_cores": dict(legs[ncpu], cores=ncpu), "e2e": e2e,
            "sample": "%s: sample_batch(256) + local_update, hidden %s, %d-row host buffer (SURVEY 8d recipe; 1M rows = minutes of first-touch page faults in this sandbox), %s warm-up + "
                      "%s timed updates in %.1f s (value = median of 3 repeats, min / max beside it) at 4 torch threads, torch %s / numpy %s CPU, %d host cores"
                      % (who, "x".join(map(str, hidden)), n_rows, v4.get("warmup"), v4.get("updates"), v4.get("seconds", 0.0),
                         torch.__version__, np.__version__, ncpu)}


def cpu_baseline(hidden, timeout_s=150):
    """runs cpu_baseline_child in its own process under a hard timeout: a CPU leg can never stall the GPU measurement"""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--hidden", ",".join(map(str, hidden))]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "steps/s", "cores": 4, "kind": "port", "sample": "CPU baseline leg exceeded %d s and was stopped" % timeout_s}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"value": None, "unit": "steps/s", "cores": 4, "kind": "port", "sample": "CPU baseline leg failed: %s" % r.stderr[-300:]}
    return json.loads(lines[-1])


E2E_ENV = "synth_humanoid"


def e2e_kwargs(hidden, batch, **over):
    """the flat argument dict of example_train/dsacv2_mlp_mujoco_offserial.py for the end-to-end loop on the
    Humanoid-SHAPED synthetic environment (tests/envs/synth_humanoid_data.py: obs 376, act 17, a table lookup per step)"""
    import numpy as np

    envs = os.path.join(ROOT, "tests", "envs")
    if envs not in sys.path:
        sys.path.append(envs)
    kw = dict(
        env_id=E2E_ENV, algorithm="DSAC_V2_HIP", seed=7, action_type="continu", reward_scale=1, is_render=False,
        value_func_name="ActionValueDistri", value_func_type="MLP", value_hidden_sizes=list(hidden),
        value_hidden_activation="gelu", value_output_activation="linear", value_min_log_std=-8, value_max_log_std=8,
        policy_func_name="StochaPolicy", policy_func_type="MLP", policy_act_distribution="TanhGaussDistribution",
        policy_hidden_sizes=list(hidden), policy_hidden_activation="gelu", policy_output_activation="linear",
        policy_min_log_std=-20, policy_max_log_std=0.5, value_learning_rate=1e-4, policy_learning_rate=1e-4,
        alpha_learning_rate=3e-4, gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, delay_update=2, TD_bound=1, bound=True,
        trainer="off_serial_trainer", ini_network_dir=None, buffer_name="hip_replay_buffer", buffer_warm_size=2000,
        buffer_max_size=100000, replay_batch_size=batch, sample_interval=1, sample_batch_size=20, batch_size_per_sampler=20,
        noise_params=None, num_eval_episode=1, eval_interval=10 ** 9, eval_save=False, save_folder=None,
        apprfunc_save_interval=10 ** 9, log_save_interval=10 ** 9, max_iteration=10 ** 9, use_gpu=True, enable_cuda=True,
        obsv_dim=O, action_dim=A, action_high_limit=np.full((A,), 0.4, np.float32),
        action_low_limit=np.full((A,), -0.4, np.float32), additional_info={}, cnn_shared=False,
    )
    kw.update(over)
    return kw


def _timed_loop(trainer, sampler, warm, iters, sync):
    """`warm` + `iters` iterations of trainer.step(); returns (seconds, sampler seconds) of the timed part"""
    acc = {"s": 0.0}
    inner = sampler.sample

    def sample():
        t0 = time.perf_counter()
        out = inner()
        acc["s"] += time.perf_counter() - t0
        return out

    sampler.sample = sample
    for _ in range(warm):
        trainer.step()
        trainer.iteration += 1
    sync()
    acc["s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(iters):
        trainer.step()
        trainer.iteration += 1
    sync()
    return time.perf_counter() - t0, acc["s"]


def e2e_gpu(hidden, device, iters=400, warm=60):
    """BASELINE.json configs[1] END TO END through the drop-in surface (training/trainer.py:60-82): per iteration the
    sampler takes 20 environment steps acting with the live learner weights (single-launch HIP forward, dsact_act.h),
    add_batch writes them to the HBM ring, sample_batch draws 256 indices (np.random.randint) and gathers, local_update
    enqueues one update. Nothing is logged or evaluated inside the timed part. Not the headline value."""
    import plugin

    kw = e2e_kwargs(hidden, B, hip_device=device)
    alg = plugin.create_alg(**kw)
    sampler = plugin.create_sampler(**kw)
    buf = plugin.create_buffer(**kw)
    trainer = plugin.create_trainer(alg, sampler, buf, None, **kw)
    w, ws = _timed_loop(trainer, sampler, warm, iters, alg.engine.sync)
    e = alg.engine
    obs1 = sampler.obs[None].astype("float32")
    for _ in range(50):
        e.policy_forward(obs1)
    t0 = time.perf_counter()
    for _ in range(500):
        e.policy_forward(obs1)
    fwd_us = (time.perf_counter() - t0) / 500 * 1e6
    try:
        if e.debug_get("act_host") == 1.0:   # the acting forward runs on the calling thread (csrc/dsact_host_act.h)
            fwd_split = {"where": "host", "forward_us": e.debug_get("act_host_us"), "threads": e.debug_get("act_host_threads"),
                         "isa": {0.0: "x86-64", 1.0: "avx2+fma", 2.0: "avx512f"}.get(e.debug_get("act_host_isa")), "helpers_pinned": e.debug_get("act_host_pinned"), "helpers_moved": e.debug_get("act_host_repins"), "snapshot_copies": e.debug_get("act_copies"),
                         "calls": e.debug_get("act_host_calls"), "last_copy_wait_us": e.debug_get("act_copy_wait_us")}
        else:
            fwd_split = {"where": "gpu", "launch_call_us": e.debug_get("act_launch_us"), "completion_spin_us": e.debug_get("act_wait_us")}
    except Exception:
        fwd_split = None
    return {"policy_forward_split": fwd_split, "value": iters / w, "unit": "iterations/s", "ms_per_iteration": 1000.0 * w / iters,
            "sampler_ms_per_iteration": 1000.0 * ws / iters, "env_steps_per_iteration": kw["sample_batch_size"],
            "policy_forward_us": fwd_us, "iterations": iters,
            "note": "HipOffSerialTrainer.step(): 20 env steps (Humanoid-shaped table-lookup env, ~2 us/step: the figure is the "
                    "framework's cost, not MuJoCo's) + add_batch + sample_batch(256) + local_update; policy_forward_us = one acting "
                    "forward call: on the host from the pinned policy snapshot refreshed behind every policy-moving update "
                    "(csrc/dsact_host_act.h; `hip_host_act=False`: one GPU launch per call, csrc/dsact_act.h)"}


def e2e_gpu_grouped(hidden, device, interval=8, iters=1600, warm=160):
    """the same loop with the reference's CNN-example cadence sample_interval = 8 (example_train/dsacv2_cnn_carracing_offasync.py:133;
    loop training/trainer.py:63-82): one sampler call, then 8 updates -- which HipOffSerialTrainer.train() issues as ONE graph
    replay (HipReplayBuffer.sample_batches + DSAC_V2_HIP.local_update_group = dsact_run_group: the pipelined graph). Reported:
    iterations/s and the cost of an update THROUGH THE PLUGIN SURFACE = (iteration time - sampler time); the same loop with
    hip_group_updates=False (one local_update call per iteration) beside it."""
    import plugin

    out = {}
    for grouped in (True, False):
        kw = e2e_kwargs(hidden, B, hip_device=device, sample_interval=interval, hip_group_updates=grouped)
        alg = plugin.create_alg(**kw)
        sampler = plugin.create_sampler(**kw)
        buf = plugin.create_buffer(**kw)
        trainer = plugin.create_trainer(alg, sampler, buf, None, **kw)
        trainer._grouping = True          # what train() sets: step() alone keeps one update per call
        w, ws = _timed_loop(trainer, sampler, warm, iters, alg.engine.sync)
        # the sampler ALONE (device idle when it starts): inside the loop its first acting forward queues behind the updates
        # in flight, so the loop's sampler time contains their device time
        alg.engine.sync()
        t0 = time.perf_counter()
        for _ in range(40):
            sampler.sample()
        s_only = (time.perf_counter() - t0) / 40
        r = {"value": iters / w, "unit": "iterations/s", "ms_per_iteration": 1000.0 * w / iters,
             "sampler_ms_per_iteration": 1000.0 * ws / iters, "sampler_alone_ms_per_call": 1000.0 * s_only,
             "update_us_through_the_surface": 1e6 * (w / iters - s_only / interval),
             "host_us_per_update": 1e6 * (w - ws) / iters}
        if grouped:
            try:
                r["pipelined_graph"] = alg.engine.debug_get("pipe_graph") == 1.0
            except Exception:
                pass
            out.update(r)
        else:
            out["ungrouped"] = r
        del trainer, buf, sampler, alg
    out.update({"sample_interval": interval, "iterations": iters,
                "note": "HipOffSerialTrainer with sample_interval = %d: per %d iterations one sampler call (20 env steps) and ONE graph "
                        "replay of %d updates; update_us_through_the_surface = iteration time - (sampler call alone) / %d: what an update "
                        "costs through sample_batches + local_update_group with everything the loop adds (host enqueue, the acting "
                        "forward waiting for the updates in flight); host_us_per_update = iteration time - sampler time inside the loop "
                        "(host work only: the device time of the updates shows up in the sampler's first acting forward)"
                        % (interval, interval, interval, interval)})
    return out


def e2e_cpu(hidden, budget_s=10.0):
    """the same loop on the host cores: the UNMODIFIED reference trainer (its own factories, CPU nets) where
    /root/reference is mounted -- kind "reference"; elsewhere our loop around the oracle port of the update and a CPU
    container for acting -- kind "port". A handful of iterations: one CPU update is ~20 ms."""
    import tempfile

    import numpy as np
    import torch
    from oracle import ref_loader

    torch.set_num_threads(4)
    kw = e2e_kwargs(hidden, B, algorithm="DSAC_V2", buffer_name="replay_buffer", use_gpu=False, enable_cuda=False,
                    buffer_warm_size=400, buffer_max_size=20000, eval_interval=10 ** 9, max_episode_steps=None)
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "apprfunc"), exist_ok=True)
    kw["save_folder"] = d
    if ref_loader.reference_available():
        from oracle.trainer_trajectory import install_writer_stub

        ref_loader.import_reference()
        install_writer_stub()
        from training.evaluator import create_evaluator
        from training.off_sampler import create_sampler
        from training.trainer import create_trainer
        from utils.initialization import create_alg, create_buffer

        torch.manual_seed(0)
        alg = create_alg(**kw)
        sampler, buf = create_sampler(**kw), create_buffer(**kw)
        ev = create_evaluator(**dict(kw, max_episode_steps=20))
        trainer = create_trainer(alg, sampler, buf, ev, **kw)
        trainer.iteration = 1          # iteration 0 would evaluate and checkpoint
        kind = "reference"
    else:
        import plugin
        from dsac_v2_hip import ApproxContainer
        from oracle.dsact_oracle import DsactOracle, ReplayOracle, default_config, draw_noise

        torch.manual_seed(0)
        orc = DsactOracle(default_config(O, A, hidden))

        class PortAlg:
            networks = ApproxContainer(**dict(kw, algorithm="DSAC_V2_HIP"))

            def local_update(self, data, it):
                return orc.local_update(data, draw_noise(B, A), it)

        class PortBuffer:
            r = ReplayOracle(O, A, kw["buffer_max_size"])
            size = property(lambda self: self.r.size)
            add_batch = lambda self, s: self.r.add_batch(s)
            sample_batch = lambda self, n: self.r.sample_batch(n)
            __get_RAM__ = lambda self: 0.0

        alg, buf = PortAlg(), PortBuffer()
        sampler = plugin.create_sampler(**dict(kw, algorithm="DSAC_V2_HIP"))
        trainer = plugin.create_trainer(alg, sampler, buf, None, **dict(kw, save_folder=None))
        trainer.iteration = 1
        kind = "port"
    t0 = time.perf_counter()
    trainer.step(); trainer.iteration += 1
    one = time.perf_counter() - t0
    iters = max(3, min(40, int(budget_s / max(one, 1e-3))))
    w, ws = _timed_loop(trainer, sampler, 1, iters, lambda: None)
    return {"value": iters / w, "unit": "iterations/s", "ms_per_iteration": 1000.0 * w / iters,
            "sampler_ms_per_iteration": 1000.0 * ws / iters, "cores": 4, "kind": kind, "iterations": iters}


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
    # the sampler side of the boundary (training/off_sampler.py:44-51,82 -> trainer.py:58-61): the batch-1 policy
    # forward that acts with the live learner weights (dsact_policy_forward: H2D of one observation, fused MLP forward,
    # D2H of the logits, synchronous) -- the device part of "Time/Sampler time" -- and add_batch of 20 transitions
    # (pinned staging, one H2D, asynchronous ring write)
    obs1 = torch.as_tensor(rng.standard_normal((1, O), dtype=np.float32))
    for _ in range(20):
        alg.networks.policy(obs1)
    t0 = time.perf_counter()
    for _ in range(200):
        alg.networks.policy(obs1)
    res["sampler_policy_forward_us"] = (time.perf_counter() - t0) / 200 * 1e6
    if e.buffer_size > 0:
        nb = 20
        cols = (rng.standard_normal((nb, O), dtype=np.float32), rng.uniform(-.4, .4, (nb, A)).astype(np.float32),
                rng.standard_normal(nb, dtype=np.float32), rng.standard_normal((nb, O), dtype=np.float32),
                np.zeros(nb, np.float32), np.zeros(nb, np.float32))
        for _ in range(10):
            e.buffer_add(*cols)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            e.buffer_add(*cols)
        res["buffer_add_20_call_us"] = (time.perf_counter() - t0) / 200 * 1e6   # host time of the call (no sync inside)
        e.sync()
        res["buffer_add_20_drained_us"] = (time.perf_counter() - t0) / 200 * 1e6
    res["bytes_per_step"] = 4 * Bt * (2 * O + A + 2)
    res["note"] = ("local_update(data) with a minibatch from outside the HIP ring, %d eager updates each: host_batch = "
                   "CPU tensors (PCIe-inclusive), cuda_batch = CUDA tensors; not the headline value" % n)
    return res


def graph_steps(steps, warmup, cap=64, even=False):
    """updates captured per hipGraph: the largest divisor (<= cap) of both the timed and the warm-up step count, so
    that exactly W and exactly K steps are replayed (measured: 2/graph -> 9,248, 8 -> 9,416, 40 -> 9,473 steps/s on the
    round-1 chain: the gap between graph launches amortises). even: the fast mode needs whole delay_update periods."""
    import math

    if os.environ.get("DSACT_BENCH_GRAPH_STEPS"):
        return int(os.environ["DSACT_BENCH_GRAPH_STEPS"])
    g = math.gcd(int(steps), int(warmup)) if warmup else int(steps)
    best = 1
    for d in range(1, cap + 1):
        if g % d == 0 and (not even or d % 2 == 0):
            best = d
    return best


def n_regions(steps):
    if os.environ.get("DSACT_BENCH_REGIONS"):   # experiments only
        return int(os.environ["DSACT_BENCH_REGIONS"])
    # short regions drift while the clocks ramp after an idle GPU (1.48 -> 1.33 ms over 15 regions at K = 20, DESIGN.md
    # section 6a): 61 regions of K <= 100 steps are still < 0.1 s of GPU time and put the median in the settled regime
    return 61 if steps <= 100 else 5 if steps <= 10000 else 3


def measure(alg, steps, warmup, world=1, dp=None, flags=0):
    """W warm-up steps, then R regions of exactly `steps` steps; returns (median region wall seconds, all regions [s],
    hipEvent ms of the median region or None). Every region: barrier + device sync, clock, K steps, device sync +
    barrier, clock; MAX over ranks."""
    import torch

    e = alg.engine
    regions, evs = [], []
    R = n_regions(steps)
    gs = graph_steps(steps, warmup, even=bool(flags & 1))
    short = steps <= 64 and not (flags & 1 and steps & 1)   # short runs: the timed region is ONE graph of exactly K steps
    if dp is None:
        if short:
            gs = steps
            if warmup:
                e.time_steps(0, warmup, use_graph=False, flags=flags)   # exactly W eager warm-up updates
            e.graph_build(gs, flags)
        else:
            e.graph_build(gs, flags)
            if warmup:
                e.graph_run(0, warmup)
        e.sync()
        it = warmup
        for _ in range(R):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ms = e.time_steps(it, steps, use_graph=True)  # hipEvents on the engine's stream + host wait for the end event
            torch.cuda.synchronize()
            regions.append(time.perf_counter() - t0)
            evs.append(ms)
            it += steps
    else:
        import torch.distributed as dist

        native = getattr(dp, "native", False)
        if native:
            if short:
                gs = steps
                e.dp_begin(0)
                for _ in range(warmup):
                    dp.step()
                native = dp.build_graph(gs)      # False: capture failed somewhere -> every rank steps eagerly (loud message)
            else:
                native = dp.build_graph(gs)
                if warmup:
                    dp.run(0, warmup)
            measure.dp_graph = native
        else:
            e.dp_begin(0)
            for _ in range(warmup):
                dp.step()
        e.sync()
        it = warmup
        for _ in range(R):
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if native:
                dp.run_graph(it, steps)
            else:
                for _ in range(steps):
                    dp.step()
            e.sync()
            torch.cuda.synchronize()
            dist.barrier()
            w = time.perf_counter() - t0
            t = torch.tensor([w], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            regions.append(float(t.item()))
            evs.append(None)
            it += steps
    order = sorted(range(R), key=lambda i: regions[i])
    mid = order[R // 2]
    measure.last_graph_steps = gs
    return regions[mid], regions, evs[mid]


def chain_flops(lay, batch):
    """algorithmic FLOP per launch of the five launches of the chain path (2 x multiply-accumulates, elementwise
    work ignored): forward A / B, loss + critic backward, policy backward (+ the critics' weight gradients riding in
    it), the policy's weight gradients."""
    F, A_, W, L = lay.obs_dim, lay.act_dim, lay.hidden[0], len(lay.hidden)
    hid = (L - 1) * W * W
    pi = F * W + hid + W * 2 * A_
    q = (F + A_) * W + hid + W * 2
    dw_q = q            # dW of a net: one MAC per parameter per sample
    dw_pi = pi
    per_row = {
        "chain_fwd_a": 2 * pi + 2 * q + 2 * F * W,
        "chain_fwd_b": 4 * (A_ * W + hid + 2 * W),
        "chain_bwd_q": 4 * (2 * W + hid) + 2 * A_ * W,
        "chain_bwd_pi": 2 * A_ * W + hid + 2 * dw_q,
        "dW": dw_pi,
    }
    per_row["chain_fwd"] = per_row["chain_fwd_a"] + per_row["chain_fwd_b"]   # merged A+B launch (batch <= 256)
    # pipelined graph (k_chain_fwdp): "+next" also runs pi(obs') and pi_target(obs2') of the next minibatch; "_q" holds only
    # the chains that need the fresh critics (its own policy units ran in the previous launch)
    per_row["chain_fwd+next"] = per_row["chain_fwd"] + 2 * pi
    per_row["chain_fwd_q"] = per_row["chain_fwd"] - 2 * pi
    per_row["chain_fwd_q+next"] = per_row["chain_fwd"]
    # the last launch of an update whose (discarded) policy backward is deferred: the critics' weight-gradient tiles only;
    # `deferred_bwd_pi` is what the next forward launch then carries on top of its own chains
    per_row["chain_dw_q"] = 2 * dw_q
    per_row["chain_bwd_qt"] = per_row["chain_bwd_q"] + 2 * dw_q    # round 5: critics' backward + their tiles + close in one launch
    per_row["chain_bwd_qpt"] = per_row["chain_bwd_q"] + per_row["chain_bwd_pi"] + dw_pi   # ... and of a policy-moving update: the whole backward
    per_row["deferred_bwd_pi"] = 2 * A_ * W + hid + dw_pi
    per_row["chain_bwd"] = per_row["chain_bwd_q"] + per_row["chain_bwd_pi"] + per_row["dW"]   # merged backward + optimiser launch
    return {k: 2.0 * v * batch for k, v in per_row.items()}


PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")


def _kernel_base_name(full):
    """'void dsact::k_chain_fwdp<4, false>(dsact::PipeFwd const*)' -> 'k_chain_fwdp'"""
    import re

    m = re.search(r"dsact::(k_\w+)", full)
    return m.group(1) if m else full


def pmc_traffic(kernel_name, pick="max"):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/rNN_pmc_traffic.json, produced
    by scripts/gpu_r*_prof.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench with --kernel-trace only,
    FETCH_SIZE doubled per the MI355X guide's gfx950 note). Not measurable live (needs rocprofv3); None if absent.
    `kernel_name` is matched EXACTLY against the kernel's base name (k_chain_fwdp does not pick up k_chain_fwdpb)."""
    d = None
    for name in PMC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))["mlp"]
            pmc_traffic.source = "profiles/" + name
            break
        except (OSError, KeyError, ValueError):
            continue
    if d is None:
        return None
    # several instantiations may match (k_chain_fwd<4, 2> = group A, <4, 1> = group B at batch 256): `pick` chooses
    best = None
    for k, v in d.items():
        if _kernel_base_name(k) == kernel_name:
            per = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
            if best is None or (per > best if pick == "max" else per < best):
                best = per
    return best


def profile_kernels(e, first_it, n_steps=12, skip=2):
    """average in-chain duration per launch name over n_steps eager updates (the dispatch's own start/stop events)"""
    acc, cnt, blocks, order = {}, {}, {}, []
    for i in range(n_steps):
        prof = e.profile_step(first_it + i)
        e.sync()
        if i < skip:
            continue
        for name, ms, b in prof:
            if name not in acc:
                acc[name], cnt[name], blocks[name] = 0.0, 0, b
                order.append(name)
            acc[name] += ms
            cnt[name] += 1
    return [(n, acc[n] / cnt[n], blocks[n]) for n in order]


def profile_kernels_pipe(e, first_it, n_steps, reps=10, skip=2):
    """the same for the PIPELINED launch sequence: `reps` eager runs of the sequence graph_build(n_steps) captures (its
    forward launches come in shapes, named by what they hold); average per launch name + launches per update"""
    acc, cnt, blocks, order = {}, {}, {}, []
    for i in range(reps):
        prof = e.profile_steps(first_it + i * n_steps, n_steps)
        e.sync()
        if i < skip:
            continue
        for name, ms, b in prof:
            if name not in acc:
                acc[name], cnt[name], blocks[name] = 0.0, 0, b
                order.append(name)
            acc[name] += ms
            cnt[name] += 1
    per_update = {n: cnt[n] / float((reps - skip) * n_steps) for n in order}
    return [(n, acc[n] / cnt[n], blocks[n]) for n in order], per_update


def self_spawn(args):
    """`python bench.py --gpus N` without a torch.distributed environment: re-execute under torch.distributed.run with N
    ranks on this node. Never measures fewer GPUs than were asked for."""
    import subprocess

    if not args.dry_run_cpu:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but only %d device(s) visible -- refusing to measure a smaller job\n" % (args.gpus, have))
            sys.exit(2)
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--hidden", type=str, default="256,256,256",
                    help="reference default (example_train/*.py); BASELINE.json words it as 256,256 -> reported in `alt`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed regions of the headline configuration and nothing else (no fast / alt / V1 / e2e / CNN legs, no eager "
                         "per-kernel profile): what a rocprofv3 pass should see so that its kernel statistics are this configuration's alone")
    ap.add_argument("--fast", action="store_true", help="primary number with DSACT_F_SKIP_ACTOR_ON_OFF_ITERS (default: strict; fast is reported in `fast`)")
    ap.add_argument("--replay-rows", "--rows", dest="replay_rows", type=int, default=N_REPLAY,
                    help="rows of the replay ring in HBM per GPU (configs[1]: 1M; configs[4]: 10M = 30.9 GB)")
    ap.add_argument("--batch", type=int, default=B, help="minibatch rows per GPU (BASELINE metric: 256)")
    ap.add_argument("--cnn-only", action="store_true", help="measure only the CNN workload (configs[3]); prints its object")
    ap.add_argument("--cnn-steps", type=int, default=400)
    ap.add_argument("--cnn-type", type=str, default="type_2", help="type_2 at (3,96,96) (default) or type_1 at (4,84,84) (SURVEY.md D3)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher check without GPUs: the ranks rendezvous over gloo, all-reduce one tensor and rank 0 prints the line")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: print the cpu_baseline object and exit")
    ap.add_argument("--dp-eager", action="store_true", help="data-parallel leg through torch.distributed eagerly instead of the graph-captured native RCCL path")
    args = ap.parse_args()
    if args.headline_only:
        args.no_alt = args.no_cpu_baseline = True
    steps, warmup = int(args.steps), int(args.warmup)
    if args.fast:   # whole delay_update periods per graph
        steps += steps & 1
        warmup += warmup & 1

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_child([int(x) for x in args.hidden.split(",")])))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    rank, world, local = env_int("RANK", 0), env_int("WORLD_SIZE", 1), env_int("LOCAL_RANK", 0)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    if args.dry_run_cpu:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
        t = torch.ones(4) * (rank + 1)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "sum": float(t[0].item()), "steps": steps, "warmup": warmup}))
        dist.destroy_process_group()
        return

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    import torch

    if torch.cuda.device_count() < max(1, world):
        sys.stderr.write("bench.py: %d rank(s) but %d device(s) visible\n" % (world, torch.cuda.device_count()))
        sys.exit(2)
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
    dp_mode = None
    if use_dp:
        from dsact.dp import DataParallelUpdater

        native = not args.dp_eager
        try:
            # native: the library's own RCCL communicator; the whole update (gather -> gradients -> all-reduce ->
            # Adam/Polyak) is captured in one hipGraph per rank
            dp = DataParallelUpdater(e, broadcast_tensors=(e.online, e.target, e.adam_m, e.adam_v), native=native,
                                     overlap=(not native) and os.environ.get("DSACT_DP_OVERLAP", "0") == "1")
        except Exception as ex:
            if not native:
                raise
            sys.stderr.write("bench.py: native RCCL communicator unavailable (%r): eager torch.distributed collectives\n" % (ex,))
            dp = DataParallelUpdater(e, broadcast_tensors=(e.online, e.target, e.adam_m, e.adam_v))
        dp.force_collective = world == 1   # forced-DP runs on one rank still issue the collective
        dp_mode = "hipGraph (gather -> gradients -> ncclAllReduce -> Adam/Polyak, library-owned RCCL communicator)" if dp.native \
            else "eager launches + torch.distributed all-reduce (%s)" % ("critics' segment overlapped" if dp.overlap else "single")
    wall, regions, ev_ms = measure(alg, steps, warmup, world, dp, flags=1 if args.fast else 0)
    if use_dp and getattr(dp, "native", False) and not getattr(measure, "dp_graph", True):
        dp_mode = "EAGER launches + ncclAllReduce on the library-owned RCCL communicator (the hipGraph capture of the data-parallel update FAILED on this runtime: see stderr)"
    stats = e.read_stats()
    finite = all(v == v and abs(v) < 1e30 for v in stats.values())
    updates_per_s = steps / wall
    value = updates_per_s * world  # batch-B gradient-step equivalents per second over the whole job
    lay = e.layout
    Bb = args.batch
    flop = lay.flop_per_step(Bb)
    byts = lay.bytes_per_step(Bb, 2)
    gs = getattr(measure, "last_graph_steps", graph_steps(steps, warmup, even=args.fast))
    chain = bool(e.chain_active)
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
            "launch": dp_mode if use_dp else "hipGraph (%d steps/graph; the next update's gather rides in the loss launch)" % gs,
            "kernels": ("row-slice fused chains + transposed-operand weight-gradient tiles (%d launches/update)%s" % (3 if Bb <= 256 else 4 if Bb <= 512 else 5 if Bb <= 1024 else 6, "; throughput-regime kernels (dsact_fat.h) for the forward launches" + (" and the backward launches" if Bb >= 4096 else "") if Bb >= 1024 else "")) if chain else "per-layer tile stages",
            "timing": "median of %d timed regions of exactly %d steps (each bracketed by barrier + device sync; max over ranks)" % (len(regions), steps),
            "replay_fill": "torch device generator, the distributions of SURVEY.md 8(d) (a host np.random.default_rng(0) fill would push "
                           "3 GB through PCIe); indices: np.random.seed(1 + rank) + np.random.randint, a %d-row table cycled" % IDX_ROWS,
            "unit_note": "value = synchronized updates/s x n_gpus (each rank contributes one batch-%d gradient per update)" % args.batch,
        },
        "regions_ms": [round(1000.0 * r, 4) for r in regions],
        "finite_stats": finite,
    }
    if rank == 0:
        per_gpu_steps = updates_per_s
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
            if args.headline_only:
                raise RuntimeError("--headline-only: no eager per-kernel profile")
            pipe = (not use_dp) and e.debug_get("pipe_graph") == 1.0
            if pipe:
                # the timed graph is the pipelined one: profile ITS launch sequence (same first-iteration parity, same length)
                prof, per_update = profile_kernels_pipe(e, warmup + steps * len(regions), gs)
                out["kernels"] = [{"name": n, "us": round(ms * 1000, 2), "blocks": b, "launches_per_update": round(per_update[n], 4)}
                                  for n, ms, b in prof]
                out["kernels_per_update_us"] = round(sum(ms * 1000.0 * per_update[n] for n, ms, _ in prof if n not in ("gather", "pack")), 2)
                lpu = sum(per_update[n] for n, _, _ in prof if n not in ("gather", "pack"))
                out["config"]["kernels"] = ("row-slice fused chains + transposed-operand weight-gradient tiles with fused Adam / Polyak: "
                                            "%.3g launches/update in the timed (pipelined) graph -- one forward launch + one merged backward launch" % lpu)
                out["config"]["launch"] = ("hipGraph (%d steps/graph), delayed-update-aware pipelining: the forward launch of an update that "
                                           "leaves the policy alone also runs pi / pi_target of the next minibatch (chain_fwd+next), the next "
                                           "update's forward holds only the fresh-critic chains (chain_fwd_q); gather rides two updates ahead" % gs)
            else:
                prof = profile_kernels(e, warmup + steps * len(regions))
                out["kernels"] = [{"name": n, "us": round(ms * 1000, 2), "blocks": b} for n, ms, b in prof]
            out["kernels_note"] = ("average in-chain duration per launch: start/stop events attached to each dispatch (hipExtLaunchKernelGGL) "
                                   "-- the kernel's own begin / end timestamps, what rocprofv3 --kernel-trace reports -- over eager runs of the "
                                   "launch sequence the timed graph captures; rocprofv3 summary of this configuration ALONE: "
                                   "profiles/r06_bench_kernel_stats.csv, two consecutive updates launch by launch: profiles/r06_step_trace.txt")
            # the two streaming parts of the update, reported separately
            kus = {n: ms * 1000.0 for n, ms, _ in prof}
            if "gather" in kus:
                gb = 4.0 * Bb * (2 * lay.obs_dim + lay.act_dim + 2)
                out["roofline_gather"] = {
                    "bound": "hbm", "bytes": gb, "avg_launch_us": kus["gather"], "achieved": gb / kus["gather"] / 1e3, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": gb / kus["gather"] / 1e3 / HBM_PEAK_GBS,
                    "note": "k_gather as its own launch (eager updates): %d random replay rows of %d B -- two dependent HBM round trips, "
                            "latency- not bandwidth-bound at this size; in the timed graph replays the gather of update s+1 rides in the "
                            "critics' backward launch of update s and costs no launch of its own" % (Bb, 4 * lay.obs_dim)}
            opt_names = [n for n in ("chain_bwd_pi", "dW", "adam_polyak") if n in kus]
            if chain and opt_names:
                n_on = 2 * lay.n_q + lay.n_pi
                ob = 24.0 * 2 * lay.n_q + (24.0 * lay.n_pi + 8.0 * n_on) / 2.0
                ous = sum(kus[n] for n in opt_names)
                out["roofline_adam"] = {
                    "bound": "hbm", "bytes": ob, "us": ous, "achieved": ob / ous / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ob / ous / 1e3 / HBM_PEAK_GBS, "launches": opt_names,
                    "note": "Adam (24 B/param: read and write p, m, v) + Polyak (8 B/param on every 2nd update) of SURVEY.md 8(d). The "
                            "optimiser has no launch of its own: it is the epilogue of the weight-gradient tiles, so `us` is the duration of "
                            "the launches that carry those tiles (which also hold the policy backward chain and the tiles' MFMA work)"}
            timed = [r for r in prof if r[0] not in ("gather", "pack")]
            dom = max(timed, key=lambda r: r[1])
            out["dominant_kernel"] = {"name": dom[0], "us": round(dom[1] * 1000, 2)}
            if chain and Bb % 256 == 0 or chain and Bb <= 256:
                fl = chain_flops(lay, Bb)
                if any(r[0] == "chain_dw_q" for r in prof):   # the policy backward of the previous update rides in these launches
                    fl["chain_fwd_q"] += fl["deferred_bwd_pi"]
                    fl["chain_fwd_q+next"] += fl["deferred_bwd_pi"]
                kname = {"chain_fwd+next": "dsact::k_chain_fwdp (own minibatch: all 8 chains; + policy / policy_target of the next minibatch)",
                         "chain_fwd_q": "dsact::k_chain_fwdp (fresh-critic chains only)", "chain_fwd_q+next": "dsact::k_chain_fwdp",
                         "chain_bwd": "dsact::k_chain_bwd2 (critics' + policy backward + all dW/Adam tiles in one launch)",
                         "chain_fwd": "dsact::k_chain_fwd2 (groups A + B in one launch)", "chain_fwd_a": "dsact::k_chain_fwd (group A)", "chain_fwd_b": "dsact::k_chain_fwd (group B)",
                         "chain_bwd_q": "dsact::k_chain_bwd_q", "chain_bwd_qt": "dsact::k_chain_bwd_qt (critics' backward + their dW/Adam tiles + close)",
                         "chain_bwd_qpt": "dsact::k_chain_bwd_qpt (critics' backward -> policy backward -> all dW/Adam/Polyak tiles -> close)", "chain_bwd_pi": "dsact::k_chain_bwd_pi (+ riding k_dw2 tiles)",
                         "dW": "dsact::k_dw2"}.get(dom[0], dom[0])
                if dom[0] in fl:
                    dur_us = dom[1] * 1000.0
                    ach = fl[dom[0]] / (dur_us * 1e-6) / 1e12
                    out["roofline"] = {
                        "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                        "traffic": pmc_traffic({"chain_bwd": "k_chain_bwd2", "chain_fwd": "k_chain_fwdp" if pipe else "k_chain_fwd2", "chain_fwd+next": "k_chain_fwdp", "chain_fwd_q": "k_chain_fwdpb" if any(r[0] == "chain_dw_q" for r in prof) else "k_chain_fwdp", "chain_fwd_q+next": "k_chain_fwdp", "chain_fwd_a": "k_chain_fwd", "chain_fwd_b": "k_chain_fwd", "chain_bwd_q": "k_chain_bwd_q", "chain_bwd_qt": "k_chain_bwd_qt", "chain_bwd_qpt": "k_chain_bwd_qpt",
                                                "chain_bwd_pi": "k_chain_bwd_pi", "dW": "k_dw2"}.get(dom[0], dom[0]),
                                               "min" if dom[0] == "chain_fwd_b" else "max"),
                        "traffic_source": "%s (a committed rocprofv3 --pmc pass of this bench; NOT measured in this run)" % getattr(pmc_traffic, "source", None),
                        "kernel": kname, "avg_launch_us": dur_us, "flop_per_launch": fl[dom[0]],
                        "note": "dominant launch of the update by in-chain duration; achieved = algorithmic FLOP of the launch (2 x MAC of "
                                "the layers its workgroups run, elementwise ignored) / its average duration (dispatch start/stop events, "
                                "10 eager updates); fp32 MFMA peak == fp32 vector peak. The launch runs %d workgroups on 256 CUs: the "
                                "chip-level fraction is bounded by that occupancy (per-workgroup phase timeline: profiles/r02_chain_*_timeline.txt); "
                                "traffic = bytes/launch (2*FETCH_SIZE + WRITE_SIZE) from the committed PMC passes, or null" % dom[2],
                    }
        except Exception as ex:  # profiling is informational
            out["kernels_error"] = str(ex)
        if "roofline" not in out:
            out["roofline"] = dict(out["roofline_step"], note="whole update (no per-kernel profile available on this path)")
    if rank == 0 and not use_dp and not args.fast and not args.headline_only:
        # same workload with the actor/alpha backward skipped on the off iterations of the delayed update: the
        # reference computes and discards those gradients (dsac_v2.py:174-186 vs :324); bitwise-identical parameter
        # trajectory (tests/test_hip_parity.py::test_skip_discarded_actor_backward_keeps_trajectory)
        fs, fw = steps + (steps & 1), warmup + (warmup & 1)
        wf, _, _ = measure(alg, fs, fw, flags=1)
        out["fast"] = {"value": fs / wf, "unit": "steps/s", "ms_per_step": 1000.0 * wf / fs,
                       "note": "DSACT_F_SKIP_ACTOR_ON_OFF_ITERS; not the headline value"}
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        try:
            out["boundary"] = boundary_rates(alg, min(max(steps, 100), 600))
        except Exception as ex:  # informational leg
            out["boundary_error"] = repr(ex)
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        try:
            out["e2e"] = e2e_gpu(hidden, local)
        except Exception as ex:  # informational leg
            out["e2e_error"] = repr(ex)
        try:
            out["e2e_si8"] = e2e_gpu_grouped(hidden, local)
        except Exception as ex:  # informational leg
            out["e2e_si8_error"] = repr(ex)
    if not use_dp and not args.no_alt and args.batch == B:
        alt_hidden = [256, 256] if hidden != [256, 256] else [256, 256, 256]
        del alg
        alg2 = make_alg(alt_hidden, local, seed=0)
        fill_replay(alg2.engine, min(args.replay_rows, 200_000), seed=100)
        upload_indices(alg2.engine, min(args.replay_rows, 200_000), IDX_ROWS, seed=1)
        w2, _, _ = measure(alg2, steps, warmup)
        l2 = alg2.engine.layout
        out["alt"] = {"hidden": alt_hidden, "value": steps / w2, "unit": "steps/s",
                      "frac_fp32": l2.flop_per_step(B) * steps / w2 / 1e12 / FP32_PEAK_TFLOPS,
                      "frac_hbm": l2.bytes_per_step(B, 2) * steps / w2 / 1e9 / HBM_PEAK_GBS}
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        # the reference's DSAC_V1 (one critic) on the chain kernels + pipelined graph, same shapes
        try:
            alg1 = make_alg(hidden, local, seed=0, v1=True)
            fill_replay(alg1.engine, min(args.replay_rows, 200_000), seed=100)
            upload_indices(alg1.engine, min(args.replay_rows, 200_000), IDX_ROWS, seed=1)
            w1, _, _ = measure(alg1, steps, warmup)
            l1 = alg1.engine.layout
            out["dsac_v1"] = {"value": steps / w1, "unit": "steps/s", "ms_per_step": 1000.0 * w1 / steps,
                              "frac_fp32": l1.flop_per_step(B) * steps / w1 / 1e12 / FP32_PEAK_TFLOPS}
            del alg1
        except Exception as ex:
            out["dsac_v1_error"] = repr(ex)
    if rank == 0 and not use_dp and not args.no_alt and args.batch == B:
        try:
            out["shapes"] = bench_shapes(local)
        except Exception as ex:  # informational leg
            out["shapes_error"] = repr(ex)
    if rank == 0 and not args.no_cpu_baseline and args.batch == B:
        try:
            out["cpu_baseline"] = cpu_baseline(hidden)
            if "e2e" in out and isinstance(out["cpu_baseline"].get("e2e"), dict):
                out["e2e"]["cpu"] = out["cpu_baseline"].pop("e2e")   # the same loop on the host cores, beside the GPU figure
        except Exception as ex:
            out["cpu_baseline_error"] = repr(ex)
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
