"""experiment: is the merged forward launch slower on its FIRST run after the weights were rewritten (cold L2: the packs were
written by the Adam tiles on other XCDs) than when it is simply repeated (weights hot in the XCDs' L2s)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dsac-v2_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import hip_kwargs
from dsac_v2_hip import DSAC_V2_HIP
O, A, B, N = 376, 17, 256, 8192
alg = DSAC_V2_HIP(**hip_kwargs(O, A, (256, 256, 256), B))
e = alg.engine
e.set_device_rng(5)
e.buffer_create(N)
g = torch.Generator(device="cuda").manual_seed(1)
e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                     torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                     (torch.rand(N, device="cuda", generator=g) < .05).float())
np.random.seed(1)
e.upload_index_table(np.random.randint(0, N, size=(8, B)))
ts = e.torch_stream
def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ts):
        a.record(ts); fn(); b.record(ts)
    b.synchronize()
    return a.elapsed_time(b) * 1000
e.dp_set_strict(True)
e.dp_begin(0)
res = {"after_update": [], "repeat1": [], "repeat2": []}
for rep in range(30):
    e.dp_grads(); e.dp_apply(); e.sync()          # a full update: the weights (and packs) are rewritten
    e.gather(np.random.randint(0, N, size=B)); e.sync()
    res["after_update"].append(timed(e.dp_forward))
    res["repeat1"].append(timed(e.dp_forward))
    res["repeat2"].append(timed(e.dp_forward))
for k, v in res.items():
    v = sorted(v)
    print("%-13s median %.2f us  min %.2f  (forward launch incl. event overhead)" % (k, v[len(v) // 2], v[0]))
