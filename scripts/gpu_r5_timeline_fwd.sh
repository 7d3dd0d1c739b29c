#!/bin/bash
# instrumented library built in the container (build/libdsact_tl.so): per-workgroup phase stamps of the pipelined forward launches,
# grouped by unit, for several observation widths (first-layer study).
# usage: gpurun -- 'OBS="376 120 760" STAGES="chain_fwd+next" bash scripts/gpu_r5_timeline_fwd.sh'   (env switches pass through)
# build it first, in the container:  mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value \
#     -DDSACT_TIMELINE -shared -fPIC -Iinclude -o build/libdsact_tl.so dsac-v2_amd/csrc/dsact_api.hip      (build/ is git-ignored and travels with gpurun)
set -u
[ -f build/libdsact_tl.so ] || { echo 'build/libdsact_tl.so is missing (see the header of this script)'; exit 1; }
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp DSACT_LIB_PATH=$PWD/build/libdsact_tl.so
mkdir -p gpurun_out
for O in ${OBS:-376}; do for st in ${STAGES:-chain_fwd+next}; do
DSACT_TIMELINE_STAGE=$st STAGE=$st OBSDIM=$O python - <<'PY'
import sys, os
sys.path[:0] = ['.', 'dsac-v2_amd', 'tests']
import numpy as np, torch
from helpers import hip_kwargs
from dsac_v2_hip import DSAC_V2_HIP
O, A, B, N = int(os.environ["OBSDIM"]), 17, int(os.environ.get("BATCH", "256")), 8192
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
    e.graph_run(1 + 4 * rep, 4)
e.sync()
full = e.debug_read("timeline").view(np.int64).reshape(1024, 16)
ok = (full[:,14] != 0) & (full[:,15] != 0)
rt = full[ok]
s_obs = -(-((O + 3) // 4) // 16) * 16
print("stage %s, obs %d (first layer: %d observation steps of 4 k): %d workgroups stamped" % (stage, O, s_obs, len(rt)))
if len(rt):
    t00 = rt[:,14].min()
    for u in sorted(set(int(v) for v in rt[:,11])):
        gq = rt[rt[:,11] == u]
        b, en = (gq[:,14]-t00)/100.0, (gq[:,15]-t00)/100.0
        name = ROLE[u-1] if 1 <= u <= len(ROLE) else "?"
        idx = [k for k in range(14) if k != 11 and (gq[:,k] != 0).all()]
        segs = ["%d->%d %.2f" % (a_, b_, np.median(gq[:,b_]-gq[:,a_]) / 2400.0) for a_, b_ in zip(idx[:-1], idx[1:])]
        l0 = np.median(gq[:,2]-gq[:,1]) if (gq[:,2] != 0).all() and (gq[:,1] != 0).all() else 0
        print("  unit %-5s %3d wgs: dur med %.2f us | first-layer observation part %.0f cycles = %.1f per step | phases %s"
              % (name, len(gq), np.median(en-b), l0, l0 / s_obs, "  ".join(segs)))
PY
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_timeline_fwd.txt
