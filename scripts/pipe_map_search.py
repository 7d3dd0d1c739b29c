#!/usr/bin/env python3
"""Placement / rows-per-workgroup search for the pipelined forward launches (speed only: every variant is bit-identical).
Each candidate = a set of environment switches read at dsact_create; one process, one engine per candidate.
usage (GPU box): python scripts/pipe_map_search.py candidates.txt     # one 'label | ENV=.. ENV=..' per line
prints: label, us/update (median of 3 x 2000-step graph replays), the forward launches' in-chain durations."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dsac-v2_amd")]
import __graft_entry__ as entry

entry.build()
import numpy as np
import torch

import bench

ROWS = int(os.environ.get("SEARCH_ROWS", 262144))
STEPS, GS = 2000, 50


def run(env):
    keys = [k for k in os.environ if k.startswith("DSACT_PIPE") or k.startswith("DSACT_NO_PIPE")]
    for k in keys:
        del os.environ[k]
    os.environ.update(env)
    alg = bench.make_alg([256, 256, 256], 0, seed=0)
    e = alg.engine
    bench.fill_replay(e, ROWS, seed=100)
    bench.upload_indices(e, ROWS, bench.IDX_ROWS, seed=1)
    e.graph_build(GS)
    e.graph_run(1, 400)
    e.sync()
    ts = []
    it = 401
    for _ in range(3):
        ts.append(e.time_steps(it, STEPS, use_graph=True) * 1000.0 / STEPS)
        it += STEPS
    acc = {}
    for r in range(6):
        for name, ms, _ in e.profile_steps(it, 4):
            if r >= 2:
                acc.setdefault(name, []).append(ms * 1000.0)
        it += 4
    e.sync()
    prof = " ".join("%s=%.2f" % (k, np.mean(v)) for k, v in acc.items() if k.startswith("chain"))
    e.close()
    del alg
    torch.cuda.empty_cache()
    return sorted(ts)[1], prof


def main():
    cands = []
    for line in open(sys.argv[1]):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        label, _, envs = line.partition("|")
        env = {}
        for tok in envs.split():
            k, _, v = tok.partition("=")
            env[k] = v
        cands.append((label.strip(), env))
    for label, env in cands:
        t0 = time.time()
        try:
            us, prof = run(env)
            print("%-28s %7.2f us  %s   (%.1fs)" % (label, us, prof, time.time() - t0), flush=True)
        except Exception as ex:
            print("%-28s FAILED %r" % (label, ex), flush=True)


if __name__ == "__main__":
    main()
