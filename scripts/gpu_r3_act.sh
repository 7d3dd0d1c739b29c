#!/bin/bash
# round 3: acting forward + end-to-end loop. usage: gpurun --timeout 900 -- 'bash scripts/gpu_r3_act.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_act; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "acting or policy_forward or trajectory or end_to_end or ping_pong" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_gpu.log | tail -25
python - <<'PY' 2>&1 | tee $OUT/acting.txt
import sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, torch
alg = bench.make_alg([256, 256, 256], 0)
e = alg.engine
obs = np.random.default_rng(0).standard_normal((1, 376)).astype(np.float32)
for label, env in (("one launch, (value, call) pair hand-over, observation in the kernel arguments, logits through mapped host memory", None),):
    for _ in range(100): e.policy_forward(obs)
    t0 = time.perf_counter()
    for _ in range(2000): e.policy_forward(obs)
    per = (time.perf_counter() - t0) / 2000 * 1e6
    print("%s: engine.policy_forward %.1f us/call (host: launch call %.1f us + completion spin %.1f us, rest = Python/ctypes)" % (label, per, e.debug_get("act_launch_us"), e.debug_get("act_wait_us")))
t = torch.as_tensor(obs)
for _ in range(100): alg.networks.policy(t)
t0 = time.perf_counter()
for _ in range(2000): alg.networks.policy(t)
print("networks.policy(torch [1,376]) as the sampler calls it: %.1f us/call" % ((time.perf_counter() - t0) / 2000 * 1e6))
obs8 = np.random.default_rng(1).standard_normal((8, 376)).astype(np.float32)
for _ in range(50): e.policy_forward(obs8)
t0 = time.perf_counter()
for _ in range(500): e.policy_forward(obs8)
print("n = 8 rows (copy + tile-stage path, what every call was in round 2): %.1f us/call" % ((time.perf_counter() - t0) / 500 * 1e6))
print("e2e", {k: v for k, v in bench.e2e_gpu([256, 256, 256], 0).items() if k != "note"})
PY
