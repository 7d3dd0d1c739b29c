#!/bin/bash
# round 3: GPU tests (+ parity report) + smoke + the driver's bench command (full line) + large-batch lines.
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r3_tests.sh ["<pytest -k expr>"]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_tests; rm -rf $OUT; mkdir -p $OUT; rm -f gpurun_out/parity_report.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -n "${1:-}" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$1" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
else
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
fi
grep -n "^FAILED\|^ERROR\|passed\|failed" $OUT/pytest_gpu.log | tail -15
cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"
tail -1 $OUT/bench_driver.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.0f  us %.2f  step_frac %.3f' % (d['value'], 1000 * d['ms_per_step'], d['roofline_step']['frac']))
print('kernels', ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', [])))
for k in ('roofline', 'roofline_gather', 'roofline_adam', 'e2e', 'boundary', 'cpu_baseline', 'fast', 'alt', 'dsac_v1', 'cnn'):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ('note', 'sample', 'kernels_note')}
    print(k, json.dumps(v)[:700])
for k in d:
    if k.endswith('_error'): print(k, d[k])
"
for b in 1024 4096; do
  timeout 300 python bench.py --steps 1000 --warmup 100 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1
  tail -1 $OUT/bench_b$b.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=$b value %.0f  us %.1f  step_frac %.3f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d['roofline_step']['frac'], ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', []))))
"
done
