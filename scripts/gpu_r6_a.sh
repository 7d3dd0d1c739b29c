#!/bin/bash
# round 6, first pass: the new tests (group surface for CNN / V1, CNN trajectory at sample_interval 8, 10M-row ring, fast-flag
# groups, host-side acting, hand-off words) + the e2e legs with the acting forward on the host and on the GPU.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r6_a.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_a; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
lscpu | grep -i "model name\|^CPU(s)\|L2\|L3" > $OUT/cpu.txt; cat $OUT/cpu.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 \
  -k "host_acting or handoff_words or cnn_and_v1 or family_group or fast_flag or fast_flag_and or cnn_si8 or ten_million or act_sample or gauss_distribution_sampling or sampler" \
  > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_new.log | tail -30
python - > $OUT/e2e.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
import __graft_entry__ as entry
entry.build()
hidden = [256, 256, 256]
for host in (True, False):
    os.environ.pop("DSACT_NO_HOST_ACT", None)
    if not host:
        os.environ["DSACT_NO_HOST_ACT"] = "1"
    r = bench.e2e_gpu(hidden, 0)
    print("host_act=%s e2e %s" % (host, json.dumps({k: v for k, v in r.items() if k != "note"})))
    r = bench.e2e_gpu_grouped(hidden, 0)
    print("host_act=%s e2e_si8 %s" % (host, json.dumps({k: v for k, v in r.items() if k != "note"})))
PY
cat $OUT/e2e.txt | cut -c1-900
