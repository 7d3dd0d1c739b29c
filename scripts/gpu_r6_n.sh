#!/bin/bash
# round 6: phase timeline of a backward launch at batch 1024 (instrumented library build/libdsact_tl.so); STAGE=chain_bwd_q|chain_bwd_pi
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp DSACT_LIB_PATH=$PWD/build/libdsact_tl.so
mkdir -p gpurun_out
for st in ${STAGE:-chain_bwd_q chain_bwd_pi}; do
DSACT_TIMELINE_STAGE=$st ST=$st python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_timeline_b1024_$st.txt
import sys, os
sys.path[:0] = ['.', 'dsac-v2_amd', 'tests']
import numpy as np, torch
from helpers import hip_kwargs
from dsac_v2_hip import DSAC_V2_HIP
O, A, B, N = 376, 17, int(os.environ.get("BATCH", "1024")), 16384
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
ok = (full[:,14] != 0) & (full[:,15] != 0)
rt = full[ok]
print("%s (batch %d): %d workgroups stamped" % (os.environ["ST"], B, len(rt)))
t00 = rt[:,14].min()
for u in sorted(set(int(v) for v in rt[:,11])):
    gq = rt[rt[:,11] == u]
    b, en = (gq[:,14]-t00)/100.0, (gq[:,15]-t00)/100.0
    print("  class %d  %3d wgs: begin med %.2f max %.2f | end med %.2f p90 %.2f max %.2f | duration med %.2f max %.2f" % (
        u, len(gq), np.median(b), b.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en-b), (en-b).max()))
    idx = [k for k in range(14) if k != 11 and k != 10 and (gq[:,k] != 0).all()]
    print("        phases (us @2.4GHz cycle stamps, median): " + "  ".join("%d->%d %.2f" % (a_, b_, np.median(gq[:,b_]-gq[:,a_]) / 2400.0) for a_, b_ in zip(idx[:-1], idx[1:])))
print("launch span (first begin -> last end): %.2f us" % ((rt[:,15].max() - t00) / 100.0))
PY
done
