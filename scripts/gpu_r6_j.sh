#!/bin/bash
# round 6: CNN -- next-tile requests pinned in front of the MFMAs (k_conv_fwd_narrow, k_conv_dx_mfma; the ReLU-mask rows of
# k_conv_dx_mfma requested with the tile) vs the previous conv objects (build/libdsact_convbase.so): CNN tests, bench A/B, kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_j; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "cnn and not end_to_end" > $OUT/pytest_cnn.log 2>&1; echo "pytest cnn rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_cnn.log | tail -15
for m in new base new2 base2; do
  if [ "${m#base}" != "$m" ]; then export DSACT_LIB_PATH=$PWD/build/libdsact_convbase.so; else unset DSACT_LIB_PATH; fi
  timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline > $OUT/bench_cnn_$m.log 2>&1; echo "cnn bench $m rc=$?"
  grep '^{"cnn"' $OUT/bench_cnn_$m.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('   %.0f steps/s  %.1f us   %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', []))))"
done
unset DSACT_LIB_PATH
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cnn -o cnn -- python bench.py --cnn-only --cnn-steps 200 --no-cpu-baseline > $OUT/rocprof_cnn.log 2>&1; echo "rocprof cnn rc=$?"
cp $(find $OUT/prof_cnn -name "*kernel_stats.csv" | head -1) $OUT/cnn_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_cnn
python - $OUT/cnn_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-70s calls=%s avg=%.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
