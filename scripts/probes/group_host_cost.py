"""Where the host time of one group of 8 updates goes through the plugin surface: sample_batches, the group
token, local_update_group's Python, dsact_run_group. usage (GPU box): python scripts/probes/group_host_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import bench
import __graft_entry__ as entry

entry.build()
import plugin

kw = bench.e2e_kwargs([256, 256, 256], 256, hip_device=0, sample_interval=8)
alg = plugin.create_alg(**kw)
sampler = plugin.create_sampler(**kw)
buf = plugin.create_buffer(**kw)
trainer = plugin.create_trainer(alg, sampler, buf, None, **kw)
e = alg.engine
it = 0
for _ in range(20):
    g = buf.sample_batches(256, 8)
    alg.local_update_group(g, it)
    it += 8
e.sync()
N = 300
acc = {"draw": 0.0, "token": 0.0, "fresh": 0.0, "keep": 0.0, "run_group": 0.0, "newtb": 0.0, "sync": 0.0}
from dsac_v2_hip import HipBatchGroup

for _ in range(N):
    t0 = time.perf_counter()
    size = buf.size
    idxs = np.stack([np.random.randint(0, size, size=256) for _ in range(8)])
    t1 = time.perf_counter()
    grp = HipBatchGroup(e, idxs)
    t2 = time.perf_counter()
    grp.check_fresh()
    t3 = time.perf_counter()
    alg._keep_previous_stats()
    t4 = time.perf_counter()
    e.run_group(it, grp.idxs, None, 0)
    t5 = time.perf_counter()
    tb = alg._new_tb(t0, 8)
    t6 = time.perf_counter()
    e.sync()
    t7 = time.perf_counter()
    it += 8
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
        acc[k] += v
print("per group of 8 updates, host us (mean of %d): " % N + "  ".join("%s %.1f" % (k, 1e6 * v / N) for k, v in acc.items()))
print("   sum without the final sync: %.1f us" % (1e6 * sum(v for k, v in acc.items() if k != "sync") / N))
