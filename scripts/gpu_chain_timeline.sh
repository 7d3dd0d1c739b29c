#!/bin/bash
# instrumented build (-DDSACT_TIMELINE): per-workgroup phase stamps of the chain kernels.
# usage: gpurun -- 'bash scripts/gpu_chain_timeline.sh chain_fwd_a chain_fwd_b chain_bwd_q chain_bwd_pi'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUTF=$PWD/gpurun_out/chain_timeline.txt
mkdir -p gpurun_out /tmp/tl
cp -r dsac-v2_amd include oracle tests __graft_entry__.py /tmp/tl/
cd /tmp/tl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -DDSACT_TIMELINE -shared -fPIC -o dsac-v2_amd/lib/libdsact.so dsac-v2_amd/csrc/dsact_api.hip || exit 1
for st in "$@"; do
DSACT_TIMELINE_STAGE=$st STAGE=$st python - <<'PY'
import sys, os
sys.path[:0] = ['/tmp/tl', '/tmp/tl/dsac-v2_amd', '/tmp/tl/tests']
import numpy as np, torch
from helpers import hip_kwargs, synth_batch
from dsac_v2_hip import DSAC_V2_HIP
alg = DSAC_V2_HIP(**hip_kwargs(376, 17, (256,256,256), 256))
e = alg.engine
d = synth_batch(np.random.default_rng(0), 256, 376, 17)
for it in range(8):
    e.load_batch(*(d[k].numpy() for k in ("obs","act","rew","obs2","done")))
    e.step(it)
e.sync()
full = e.debug_read("timeline").view(np.int64).reshape(512, 16)
rt = full[(full[:,14] != 0) & (full[:,15] != 0)]
if len(rt):
    # chip-wide 100 MHz stamps (slots 14/15): chain workgroups have cycle stamps too, riders only these
    t00 = rt[:,14].min()
    # merged forward launch: group-B workgroups are the ones whose first cycle stamp after 0 is 1 then 3 (no stamp 2)
    isB = (rt[:,0] != 0) & (rt[:,2] == 0) & (rt[:,3] != 0)
    groups_rt = (("chain workgroups", (rt[:,0] != 0) & ~isB), ("group-B chain workgroups (merged launch)", isB), ("rider tiles", rt[:,0] == 0))
    for nm, sel in groups_rt:
        g = rt[sel]
        if len(g):
            b, en = (g[:,14]-t00)/100.0, (g[:,15]-t00)/100.0
            print("  realtime, %d %s: begin median %.2f max %.2f us; end median %.2f p90 %.2f max %.2f us; duration median %.2f max %.2f us"
                  % (len(g), nm, np.median(b), b.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en-b), (en-b).max()))
raw = full[:512].copy()
raw[:,14:] = 0
raw = raw[raw[:,0] != 0]
print("stage %s: %d workgroups stamped" % (os.environ["STAGE"], len(raw)))
t0 = raw[:,0].min()
for b in range(len(raw)):
    pass
names = {k: "" for k in range(16)}
# group by pattern of present stamps (unit types differ)
import collections
groups = collections.defaultdict(list)
for row in raw:
    key = tuple(int(v != 0) for v in row)
    groups[key].append(row)
for key, rows in groups.items():
    rows = np.array(rows)
    idx = [k for k in range(16) if key[k]]
    print("  %d workgroups with stamps %s: start skew (cycles after first WG) median %d max %d" % (len(rows), idx, np.median(rows[:,0]-t0), (rows[:,0]-t0).max()))
    for a, b in zip(idx[:-1], idx[1:]):
        seg = rows[:,b]-rows[:,a]
        print("     stamp %2d -> %2d : min %6d  median %6d  max %6d cycles  (%.2f us @2.4GHz)" % (a, b, seg.min(), np.median(seg), seg.max(), np.median(seg)/2400.0))
    tot = rows[:,idx[-1]]-rows[:,idx[0]]
    print("     total            : min %6d  median %6d  max %6d cycles  (%.2f us)" % (tot.min(), np.median(tot), tot.max(), np.median(tot)/2400.0))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUTF
