#!/bin/bash
# LDS swizzle check: tile-path parity tests + CNN tests + CNN bench + V1 bench. usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r3_swz.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_swz; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "cnn or ragged or v1 or large_batch or golden or b512 or dp_" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline > $OUT/bench_cnn.log 2>&1
grep "^{" $OUT/bench_cnn.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); d = d.get('cnn', d)
print('cnn steps/s %.0f  us %.1f' % (d['value'], 1000 * d['ms_per_step']))
print('   ' + ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', [])))"
