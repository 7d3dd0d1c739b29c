#!/bin/bash
# round 6: WHERE do the workgroups of the throughput-regime forward (batch 1024, group A) run? instrumented library (build/libdsact_tl.so,
# python scripts/build_variant.py tl -DDSACT_TIMELINE) stamping (XCC_ID, HW_ID) per workgroup: workgroups per CU, who shares a CU
# with whom, first-layer time of sharing and non-sharing workgroups
set -u
[ -f build/libdsact_tl.so ] || { echo 'build/libdsact_tl.so is missing'; exit 1; }
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp DSACT_LIB_PATH=$PWD/build/libdsact_tl.so
mkdir -p gpurun_out
for st in chain_fwd_a; do
DSACT_TIMELINE_STAGE=$st ST=$st python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_placement_$st.txt
import sys, os, collections
sys.path[:0] = ['.', 'dsac-v2_amd', 'tests']
import numpy as np, torch
from helpers import hip_kwargs
from dsac_v2_hip import DSAC_V2_HIP
O, A, B, N = 376, 17, 1024, 16384
alg = DSAC_V2_HIP(**hip_kwargs(O, A, (256,256,256), B))
e = alg.engine
e.set_device_rng(5)
e.buffer_create(N)
g = torch.Generator(device="cuda").manual_seed(1)
e.buffer_fill_device(0, torch.randn(N, O, device="cuda", generator=g), torch.rand(N, A, device="cuda", generator=g) - .5,
                     torch.randn(N, device="cuda", generator=g), torch.randn(N, O, device="cuda", generator=g),
                     (torch.rand(N, device="cuda", generator=g) < .05).float())
np.random.seed(1)
e.upload_index_table(np.random.randint(0, N, size=(8, B)))
e.graph_build(4)
for rep in range(6):
    e.graph_run(1 + 4 * rep, 4)
e.sync()
full = e.debug_read("timeline").view(np.int64).reshape(1024, 16)
ok = (full[:,14] != 0) & (full[:,11] != 0)
rt = full[ok]
print("%s: %d workgroups stamped" % (os.environ["ST"], len(rt)))
hw = rt[:,10]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7
key = xcc * 100000 + se * 1000 + sh * 100 + cu
per = collections.defaultdict(list)
for k, r in zip(key, rt): per[int(k)].append(r)
print("distinct CUs used: %d; workgroups per CU histogram: %s" % (len(per), dict(collections.Counter(len(v) for v in per.values()))))
print("per XCC: %s" % dict(collections.Counter(int(x) for x in xcc)))
combos = collections.Counter(tuple(sorted(int(r[11]) - 1 for r in v)) for v in per.values())
print("unit combinations per CU: %s" % dict(combos))
t00 = rt[:,14].min()
share = {id(r): len(per[int(k)]) for k, r in zip(key, rt)}
for u in sorted(set(int(v) for v in rt[:,11])):
    for nshare in (1, 2, 3):
        gq = np.array([r for k, r in zip(key, rt) if int(r[11]) == u and len(per[int(k)]) == nshare])
        if len(gq) == 0: continue
        l0 = (gq[:,2] - gq[:,1]) / 2400.0
        en = (gq[:,15] - t00) / 100.0 if (gq[:,15] != 0).all() else np.zeros(len(gq))
        print("  unit %d, %d workgroup(s) on its CU: %3d wgs  first layer (obs part) med %.2f min %.2f max %.2f us | end med %.2f min %.2f max %.2f" % (
            u - 1, nshare, len(gq), np.median(l0), l0.min(), l0.max(), np.median(en), en.min(), en.max()))
print("launch span: %.2f us" % ((rt[:,15].max() - t00) / 100.0))
PY
done
