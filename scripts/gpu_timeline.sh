#!/bin/bash
# instrumented build: per-block phase timeline of tile stages. usage: gpurun -- 'bash scripts/gpu_timeline.sh fwdA_l1 ...'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUTF=$PWD/gpurun_out/timeline.txt
mkdir -p gpurun_out /tmp/tl
cp -r dsac-v2_amd include oracle tests __graft_entry__.py /tmp/tl/
cd /tmp/tl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -DDSACT_TIMELINE -shared -fPIC -o dsac-v2_amd/lib/libdsact.so dsac-v2_amd/csrc/dsact_api.hip || exit 1
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
for it in range(int(os.environ.get('ITS', '6'))):
    e.load_batch(*(d[k].numpy() for k in ("obs","act","rew","obs2","done")))
    e.step(it)
e.sync()
raw = e.debug_read("timeline").view(np.int64).reshape(-1, 8)[:512]
raw = raw[raw[:,0] != 0]
wall = raw[:,7]-raw[:,6]
raw = raw.copy(); raw[:,6:] = 0
ns = int((raw[0] != 0).sum())
dur = raw[:,ns-1]-raw[:,0]
print("stage %s: %d blocks, %d stamps; per-block duration min %d median %d max %d cycles" % (os.environ["STAGE"], len(raw), ns, dur.min(), np.median(dur), dur.max()))
d0 = (raw[:,ns-1]-raw[:,0]).astype(float); w = wall.astype(float)
print("   wall-clock (100 MHz) ticks per block: median %d -> %.2f us ; shader clock ~ %.2f GHz ; kernel span (first start -> last end) %.2f us" % (np.median(w), np.median(w)/100.0, np.median(d0)/np.median(w)*0.1, (e.debug_read("timeline").view(np.int64).reshape(-1,8)[:len(raw),7].max() - e.debug_read("timeline").view(np.int64).reshape(-1,8)[:len(raw),6].min())/100.0))
for k in range(1, ns):
    seg = raw[:,k]-raw[:,k-1]; print("   seg %d: min %6d median %6d  max %6d" % (k, seg.min(), np.median(seg), seg.max()))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUTF
