#!/bin/bash
# full GPU test suite + driver-style and long bench; extra env A/B legs: each arg "NAME=VALUE" reruns the two bench lines with it
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/parity_report.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest_gpu.log | cut -c1-300
fi
line() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric')][-1])
print("  %.0f steps/s  %.2f us/step  ev %.2f  kernels %s" % (d["value"], d["ms_per_step"]*1e3, (d.get("hip_event_ms_per_step") or 0)*1e3, " ".join("%s=%.1f" % (k["name"].replace("chain_",""), k["us"]) for k in d.get("kernels", []))))
PY
}
run() { # tag, env
  env $2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_driver_$1.log 2>&1; echo "[$1] driver-style rc=$?"; line $OUT/bench_driver_$1.log
  env $2 timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/bench_long_$1.log 2>&1; echo "[$1] long rc=$?"; line $OUT/bench_long_$1.log
}
run base "X_=1"
i=0
for kv in "$@"; do i=$((i+1)); run "ab$i" "$kv"; done
