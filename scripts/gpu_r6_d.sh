#!/bin/bash
# round 6: host-side acting -- helpers pinned inside the calling thread's core complex vs one thread; tests of the acting paths
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_d; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 \
  -k "host_acting or handoff_words or act_sample or sampler or cnn_si8 or pipelined_graph_equals" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest.log | tail -12
for T in auto 1 2 4 8; do
python - "$T" >> $OUT/e2e.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
T = sys.argv[1]
if T != "auto":
    os.environ["DSACT_HOST_ACT_THREADS"] = T
import bench
import __graft_entry__ as entry
entry.build()
r = bench.e2e_gpu([256, 256, 256], 0)
print("threads=%s e2e %s" % (T, json.dumps({k: v for k, v in r.items() if k != "note"})))
if T in ("auto", "1"):
    r = bench.e2e_gpu_grouped([256, 256, 256], 0)
    print("threads=%s e2e_si8 %s" % (T, json.dumps({k: v for k, v in r.items() if k not in ("note", "ungrouped")})))
PY
done
cut -c1-700 $OUT/e2e.txt
