"""What the zero-padded storage buys (round 6): the same nets on the row-slice chains (stored padded, hip_pad_widths=True) and on the
tile-stage kernels (exact layout), graph replays of 2,000 updates each on a synthetic ring. usage (GPU box):
python scripts/probes/padded_widths_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import __graft_entry__ as entry

entry.build()
sys.path.insert(0, os.path.join(ROOT, "dsac-v2_amd"))
from dsac_v2_hip import DSAC_V2_HIP
from helpers import hip_kwargs

CASES = [
    ("Humanoid critics 3x256, policy 3x128", 376, 17, (256, 256, 256), (128, 128, 128), 256),
    ("Humanoid 3x200", 376, 17, (200, 200, 200), None, 256),
    ("Humanoid 2x(400->no pad: control)", 376, 17, (400, 300), None, 256),
    ("obs 24 act 6 (96, 40)", 24, 6, (96, 40), None, 256),
    ("Humanoid 3x200, batch 1024", 376, 17, (200, 200, 200), None, 1024),
]
N = 20000
for title, O, A, hv, hp, B in CASES:
    row = []
    for pad in (True, False):
        over = {"policy_hidden_sizes": list(hp)} if hp else {}
        torch.manual_seed(0)
        alg = DSAC_V2_HIP(**hip_kwargs(O, A, hv, B, hip_pad_widths=pad, **over))
        e = alg.engine
        e.set_device_rng(1)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(64, B)))
        e.graph_build(8)
        e.time_steps(1, 400, use_graph=True)
        ms = min(e.time_steps(1 + 400 * (k + 1), 2000, use_graph=True) for k in range(3))
        row.append((pad, e.layout.pad_to, e.chain_active, 2000.0 / ms * 1000.0, ms / 2000.0 * 1000.0))
        e.sync()
        e.close()
    print("%-40s" % title + "".join("  | pad_widths=%-5s stored %-4s chains %-5s %8.0f steps/s %7.2f us" % r for r in row))

# ---- observation widths that are no multiple of 4: the row-slice chains (round 6) against the tile stages (rounds 2-5: DSACT_NO_CHAIN) ----
print()
for title, O, A, hv, B in (("HalfCheetah / Walker2d obs 17 act 6, 3x256", 17, 6, (256, 256, 256), 256), ("Hopper obs 11 act 3, 3x256", 11, 3, (256, 256, 256), 256),
                           ("Ant obs 105 act 8, 3x256", 105, 8, (256, 256, 256), 256), ("Pendulum obs 3 act 1, 2x64", 3, 1, (64, 64), 256)):
    row = []
    for tiles in (False, True):
        if tiles:
            os.environ["DSACT_NO_CHAIN"] = "1"
        torch.manual_seed(0)
        alg = DSAC_V2_HIP(**hip_kwargs(O, A, hv, B))
        os.environ.pop("DSACT_NO_CHAIN", None)
        e = alg.engine
        e.set_device_rng(1)
        e.buffer_create(N)
        g = torch.Generator(device="cuda").manual_seed(1)
        e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                             torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                             (torch.rand(N, device="cuda", generator=g) < .05).float())
        np.random.seed(1)
        e.upload_index_table(np.random.randint(0, N, size=(64, B)))
        e.graph_build(8)
        e.time_steps(1, 400, use_graph=True)
        ms = min(e.time_steps(1 + 400 * (k + 1), 2000, use_graph=True) for k in range(3))
        row.append((e.chain_active, 2000.0 / ms * 1000.0, ms / 2000.0 * 1000.0))
        e.sync()
        e.close()
    print("%-46s" % title + "".join("  | chains %-5s %8.0f steps/s %7.2f us" % r for r in row))
