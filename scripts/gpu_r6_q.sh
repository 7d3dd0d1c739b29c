#!/bin/bash
# round 6: k_conv_dw's workgroups of one pixel chunk on ONE XCD (DSACT_CONV_DW_XCD=0 = block order, rounds 3-5): CNN tests, bench A/B, PMC fetch per launch
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_q; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "cnn and not end_to_end" > $OUT/pytest_cnn.log 2>&1; echo "pytest cnn rc=$?"; tail -2 $OUT/pytest_cnn.log
cnn() { echo "== $1"; env $2 timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline 2>&1 | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('   %.0f steps/s  %.1f us   %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', []) if 'conv_dw' in k['name'])))"; }
{ cnn xcd "X=1"; cnn order "DSACT_CONV_DW_XCD=0"; cnn xcd2 "X=1"; cnn order2 "DSACT_CONV_DW_XCD=0"; } 2>&1 | tee $OUT/ab.txt
for m in xcd order; do
  if [ $m = order ]; then export DSACT_CONV_DW_XCD=0; else unset DSACT_CONV_DW_XCD; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$m -o pmc -- python bench.py --cnn-only --cnn-steps 40 --no-cpu-baseline > $OUT/pmc_$m.log 2>&1
  python - $OUT/pmc_$m $m <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_conv_dw<" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
for k in acc: print("%s: %s  launches %d  fetch per launch %.1f MB (FETCH_SIZE x 2 x 1024 B)" % (sys.argv[2], k[:50], cnt[k], 2 * acc[k] / cnt[k] * 1024 / 1e6))
PY
  rm -rf $OUT/pmc_$m
done 2>&1 | tee -a $OUT/ab.txt
