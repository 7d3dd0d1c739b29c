"""Diagnostic (GPU): where the largest policy-backward difference of a parity case sits. Usage:
python scripts/diag_case.py VALUE_ACT POLICY_ACT  (ragged O=11 A=3 (96,40) B=50, first update)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dsac-v2_amd")]
import test_hip_parity as T  # noqa: E402
from helpers import synth_batch  # noqa: E402
from oracle.dsact_oracle import draw_noise  # noqa: E402

va, pa = sys.argv[1], sys.argv[2]
O, A, hid, B = 11, 3, (96, 40), 50
alg, orc = T.make_pair(O, A, hid, B, value_hidden_activation=va, policy_hidden_activation=pa)
e, L, cfg = alg.engine, len(hid), orc.cfg
rng = np.random.default_rng(5)
data = synth_batch(rng, B, O, A, lim=0.4, p_done=0.05)
torch.manual_seed(1000)
noise = draw_noise(B, A)
e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
e.compute_grads(0)
e.sync()
if cfg["value_act"] in T.KINKED or cfg["policy_act"] in T.KINKED:
    orc.act_sides = T.hip_act_sides(e, cfg, L, B)
orc.compute_gradient(data, noise, keep=True)
print("kinks", orc.act_kinks)
I = orc.inter
for l in range(L):
    got = e.debug_read("dZ.pi.%d" % l).reshape(B, -1)
    want = I["dz_pi"][l].detach().numpy().reshape(B, -1)
    z = I["z_pi"][l].detach().numpy().reshape(B, -1)
    h = e.debug_read("H.pi.%d" % l).reshape(B, -1)
    gd = e.debug_read("G.pi.%d" % l).reshape(B, -1)
    err = np.abs(got - want)
    print("layer", l, "max err", err.max(), "scale", np.abs(want).max())
    for f in np.argsort(err.reshape(-1))[::-1][:6]:
        r, u = divmod(int(f), err.shape[1])
        print("  row %d unit %d: got %.6e want %.6e  z %.6e  H %.6e  G %.6e" % (r, u, got[r, u], want[r, u], z[r, u], h[r, u], gd[r, u]))
    rows = np.argsort(err.max(axis=1))[::-1][:3]
    print("  worst rows", rows, err.max(axis=1)[rows])
lg = e.debug_read("logits_pi").reshape(B, 2 * A)
print("oracle (mean, std)", I["logits"].numpy()[rows])
print("new_act (oracle) worst rows:", I["new_act"].detach().numpy()[rows])
print("eps", noise["eps_new"].numpy()[rows])
print("logits", lg[rows])
