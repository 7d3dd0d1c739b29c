#!/bin/bash
# instrumented build (-DDSACT_TIMELINE): chip-wide stamps of the merged critic-backward launch k_chain_bwd_qpt, grouped by role
# usage: gpurun -- 'bash scripts/gpu_r5_timeline_bqpt.sh'   (env switches pass through)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUTF=$PWD/gpurun_out/r5_timeline_bqpt.txt
mkdir -p gpurun_out /tmp/tl
cp -r dsac-v2_amd include oracle tests __graft_entry__.py /tmp/tl/
cd /tmp/tl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -DDSACT_TIMELINE -shared -fPIC -o dsac-v2_amd/lib/libdsact.so dsac-v2_amd/csrc/dsact_api.hip || exit 1
DSACT_TIMELINE_STAGE=chain_bwd_qpt python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUTF
import sys, os
sys.path[:0] = ['/tmp/tl', '/tmp/tl/dsac-v2_amd', '/tmp/tl/tests']
import numpy as np, torch
from helpers import hip_kwargs
from dsac_v2_hip import DSAC_V2_HIP
O, A, B, N = 376, 17, 256, 8192
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
print("chain_bwd_qpt: %d workgroups stamped" % len(rt))
t00 = rt[(rt[:,11] >= 1) & (rt[:,11] <= 4)][:,14].min()   # launch start = first critic chain slice
names = {1: "q1c", 2: "q2c", 3: "q1p", 4: "q2p", 5: "pi chain", 10: "critic tiles", 11: "policy tiles"}
for u in sorted(set(int(v) for v in rt[:,11])):
    gq = rt[rt[:,11] == u]
    b, en = (gq[:,14]-t00)/100.0, (gq[:,15]-t00)/100.0
    line = "  %-13s %3d wgs: begin med %.2f p90 %.2f max %.2f | end med %.2f p90 %.2f max %.2f | duration med %.2f max %.2f" % (
        names.get(u, str(u)), len(gq), np.median(b), np.percentile(b, 90), b.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en-b), (en-b).max())
    if u >= 10:
        w = (gq[:,13]-t00)/100.0
        line += " | wait ended med %.2f p90 %.2f max %.2f -> after-wait med %.2f max %.2f" % (np.median(w), np.percentile(w, 90), w.max(), np.median(en-w), (en-w).max())
    print(line)
    if u < 10:
        idx = [k for k in range(13) if k != 11 and (gq[:,k] != 0).all()]
        print("        phases (us @2.4GHz cycle stamps, median): " + "  ".join("%d->%d %.2f" % (a_, b_, np.median(gq[:,b_]-gq[:,a_]) / 2400.0) for a_, b_ in zip(idx[:-1], idx[1:])))
print("launch span (first begin -> last end): %.2f us" % ((rt[:,15].max() - t00) / 100.0))
PY
