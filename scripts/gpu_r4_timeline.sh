#!/bin/bash
# instrumented build (-DDSACT_TIMELINE): per-workgroup phase stamps of the PIPELINED forward launches, grouped by unit.
# usage: gpurun -- 'bash scripts/gpu_r4_timeline.sh chain_fwd+next chain_fwd_q'   (env switches pass through)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUTF=$PWD/gpurun_out/r4_timeline.txt
mkdir -p gpurun_out /tmp/tl
cp -r dsac-v2_amd include oracle tests __graft_entry__.py /tmp/tl/
cd /tmp/tl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -DDSACT_TIMELINE -shared -fPIC -o dsac-v2_amd/lib/libdsact.so dsac-v2_amd/csrc/dsact_api.hip || exit 1
for st in "$@"; do
DSACT_TIMELINE_STAGE=$st STAGE=$st python - <<'PY'
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
stage = os.environ["STAGE"]
ROLE = ["pi", "pit", "q1c", "q2c", "pin", "pitn", "q1p", "q2p", "q1t", "q2t", "q1tn", "q2tn"]
e.graph_build(4)
for rep in range(6):
    e.graph_run(1 + 4 * rep, 4)      # odd start: (F,T) (T,F) (F,T) (T,F)
e.sync()
# the last launch of the named shape wrote the buffer; a 3-update sequence ending with it: (F,T) last is impossible -> use the graph,
# whose LAST launch of that shape stays (each launch of the shape overwrites the same slots)
full = e.debug_read("timeline").view(np.int64).reshape(512, 16)
ok = (full[:,14] != 0) & (full[:,15] != 0)
rt = full[ok]
print("stage %s: %d workgroups stamped" % (stage, len(rt)))
if len(rt):
    t00 = rt[:,14].min()
    for u in sorted(set(int(v) for v in rt[:,11])):
        gq = rt[rt[:,11] == u]
        b, en = (gq[:,14]-t00)/100.0, (gq[:,15]-t00)/100.0
        name = ROLE[u-1] if 1 <= u <= len(ROLE) else "?"
        print("  unit %-5s %3d wgs: begin med %.2f max %.2f | end med %.2f p90 %.2f max %.2f | duration med %.2f max %.2f us (100 MHz chip-wide stamps)"
              % (name, len(gq), np.median(b), b.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en-b), (en-b).max()))
        idx = [k for k in range(14) if k != 11 and (gq[:,k] != 0).all()]
        segs = []
        for a_, b_ in zip(idx[:-1], idx[1:]):
            segs.append("%d->%d %.2f" % (a_, b_, np.median(gq[:,b_]-gq[:,a_]) / 2400.0))
        print("        phases (us @2.4GHz cycle stamps, median): " + "  ".join(segs))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUTF
