#!/bin/bash
# round 5, first GPU pass: hand-over probe, the new group / gate / fast-mode tests + the trajectory tests, quick bench legs.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r5_a.sh ["<pytest -k expr>"]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r5_a; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
bash scripts/gpu_r5_ubench_handover.sh > $OUT/ubench.log 2>&1; echo "ubench rc=$?"; cp gpurun_out/r5_ubench_xcd_handover.txt $OUT/ 2>/dev/null
K="${1:-}"
timeout 1000 python -m pytest tests/test_hip_groups.py tests/test_trainer_trajectory.py -q -m gpu -p no:cacheprovider --timeout 600 ${K:+-k "$K"} > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_new.log | tail -40
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "handover or pipelined or act_sample or skip_discarded or graph_replay or sidecar" > $OUT/pytest_old.log 2>&1; echo "pytest old subset rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_old.log | tail -20
summ() { grep '^{"metric"' $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
    for k in ('fast', 'e2e'):
        if k in d: print('   %s %s' % (k, json.dumps({a: b for a, b in d[k].items() if a in ('value', 'ms_per_step', 'ms_per_iteration', 'sampler_ms_per_iteration', 'update_us_through_the_surface', 'groups')})))
except Exception as e:
    print('   parse error', e)
"; }
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_driver.log 2>&1; echo "driver command rc=$?"; summ $OUT/bench_driver.log
timeout 400 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_long.log 2>&1; echo "long rc=$?"; summ $OUT/bench_long.log
