#!/bin/bash
# round 6: host-side acting inside the driver's full bench line (other legs run first in the same process) + host cost of a group
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_e; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "host_acting or act_sample or sampler" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/pytest.log | tail -2
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_$i.log 2>&1; echo "driver command run $i rc=$?"
grep '^{"metric"' $OUT/bench_driver_$i.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f' % d['value'], ' e2e %.0f it/s' % d['e2e']['value'], json.dumps(d['e2e']['policy_forward_split']), ' e2e_si8 %.0f' % d['e2e_si8']['value'], ' surface %.1f us' % d['e2e_si8']['update_us_through_the_surface'])"
done
timeout 300 python scripts/probes/group_host_cost.py > $OUT/group_host_cost.txt 2>&1; tail -3 $OUT/group_host_cost.txt
