import sys
sys.path[:0] = ['/root/repo', '/root/repo/dsac-v2_amd', '/root/repo/tests']
import numpy as np, torch
from helpers import hip_kwargs, synth_batch
from test_hip_parity import make_pair
from oracle.dsact_oracle import draw_noise
alg, orc = make_pair(376, 17, (256,256,256), 256)
e = alg.engine
d = synth_batch(np.random.default_rng(5), 256, 376, 17, p_done=0.05)
torch.manual_seed(1000); noise = draw_noise(256, 17)
orc.compute_gradient(d, noise, keep=True)
e.load_batch(*(d[k].numpy() for k in ("obs","act","rew","obs2","done")))
e.set_noise(noise["eps_new"].numpy(), noise["eps_2"].numpy(), noise["z5"].numpy(), noise["z6"].numpy())
e.compute_grads(0); e.sync()
z = orc.inter["z_q2"][0].numpy().reshape(-1)
h = e.debug_read("H.q2c.0"); g = e.debug_read("G.q2c.0")
zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
ht = torch.nn.functional.gelu(zt); ht.sum().backward()
eh = np.abs(h - ht.detach().numpy()); eg = np.abs(g - zt.grad.numpy())
i = np.argsort(-eh)[:5]
print("worst H:", [(float(z[k]), float(h[k]), float(ht[k]), float(eh[k])) for k in i])
i = np.argsort(-eg)[:3]
print("worst G:", [(float(z[k]), float(g[k]), float(zt.grad[k]), float(eg[k])) for k in i])
print("H err vs |z| bins:", [(lo, float(eh[(np.abs(z)>=lo)&(np.abs(z)<lo+1)].max()) if ((np.abs(z)>=lo)&(np.abs(z)<lo+1)).any() else None) for lo in range(0,6)])
