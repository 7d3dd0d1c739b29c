"""GPU probe: per-layer conv activations / gradients of the HIP path vs torch autograd (CNN nets)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dsac-v2_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_cnn_parity import make_pair
from oracle.dsact_oracle import draw_noise
from oracle.dsact_oracle_cnn import synth_image_batch
import torch.nn.functional as F

obs_shape, A, ct, B = (4, 84, 84), 2, "type_1", 4
if len(sys.argv) > 1 and sys.argv[1] == "t2":
    obs_shape, A, ct, B = (3, 96, 96), 3, "type_2", 8
alg, orc, cfg = make_pair(obs_shape, A, ct, B)
e = alg.engine
for it in range(2):
    data = synth_image_batch(cfg, B, seed=it)
    torch.manual_seed(1000 + it)
    noise = draw_noise(B, A)
    # oracle with conv activations retained for every differentiated stack
    acts = {}
    orig = orc._pi, orc._q
    def conv_keep(tag, x, params, strides):
        h = x; out = []
        for j, s in enumerate(strides):
            h = F.relu(F.conv2d(h, params[2 * j], params[2 * j + 1], stride=s))
            if h.requires_grad:
                h.retain_grad()
            out.append(h)
        acts.setdefault(tag, out)
        return h.reshape(h.shape[0], -1)
    import oracle.dsact_oracle_cnn as oc
    real = oc.conv_forward
    calls = {"n": 0}
    def patched(x, params, strides, collect=None):
        calls["n"] += 1
        tag = {1: "pi", 2: "pit", 3: "q1", 4: "q2"}.get(calls["n"], "other%d" % calls["n"])
        return conv_keep(tag, x, params, strides)
    oc.conv_forward = patched
    orc.compute_gradient(data, noise)
    oc.conv_forward = real
    e.load_batch(*(data[k].numpy() for k in ("obs", "act", "rew", "obs2", "done")))
    e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
    e.compute_grads(it)
    e.sync()
    for st, tag in ((0, "q1"), (1, "q2"), (2, "pi")):
        for j, a in enumerate(acts[tag]):
            nhwc = a.detach().permute(0, 2, 3, 1).reshape(-1).numpy()
            got = e.debug_read("cact.%d.%d" % (st, j))
            ga = a.grad.permute(0, 2, 3, 1).reshape(-1).numpy()
            gg = e.debug_read("cdy.%d.%d" % (st, j))
            # torch's grad w.r.t. the post-ReLU output vs ours (already masked by relu'): mask torch's
            gm = ga * (nhwc > 0)
            print("it%d %s l%d act err %.3e (scale %.3e)  dY err %.3e (scale %.3e)  nbad %d / %d" % (
                it, tag, j, np.abs(got - nhwc).max(), np.abs(nhwc).max(), np.abs(gg - gm).max(), np.abs(gm).max(),
                int((np.abs(gg - gm) > 1e-6 * max(np.abs(gm).max(), 1e-30) + 1e-12).sum()), gm.size))
    orc.update(it); e.apply_update(it)
