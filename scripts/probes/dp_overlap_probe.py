"""Repeats the overlapped data-parallel step (world 1, gloo) against the plain one and reports where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dsac-v2_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import torch.distributed as dist
from test_hip_parity import _replay_pair
from dsact.dp import DataParallelUpdater

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)

def run(overlap, steps=6, mode="dp"):
    a = _replay_pair(17, 4, (64, 64), 64, 4096, seed=4)
    e = a.engine
    if mode == "graph":
        e.graph_build(2); e.graph_run(0, steps); e.sync()
        return a
    dp = DataParallelUpdater(e, broadcast_tensors=(e.online, e.target, e.adam_m, e.adam_v), overlap=overlap)
    dp.force_collective = True
    e.dp_begin(0)
    for _ in range(steps):
        dp.step()
    torch.cuda.synchronize()
    return a

ref = run(False)
n_c = ref.engine.critic_grad_count
bad = {"plain": 0, "overlap": 0, "graph": 0}
N = int(os.environ.get("PROBE_N", "12"))
for trial in range(N):
    for name, kw in (("plain", dict(overlap=False)), ("overlap", dict(overlap=True)), ("graph", dict(overlap=False, mode="graph"))):
        a = run(**kw)
        d = (a.engine.online != ref.engine.online).nonzero().flatten()
        if d.numel():
            bad[name] += 1
            print(trial, name, "differs at", d.numel(), "elements; first", d[:6].tolist(), "n_c", n_c, "n_online", ref.engine.online.numel(),
                  "max|d|", float((a.engine.online - ref.engine.online).abs().max()))
print("mismatching runs out of", N, ":", bad)
