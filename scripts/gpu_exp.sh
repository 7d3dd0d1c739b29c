#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail gpurun_out/build.log; exit 1; }
run() { echo "== $1"; shift; env "$@" python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  steps/s %.0f  us/step %.1f' % (d['value'], d['ms_per_step']*1000))"; }
run baseline A=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run debug_hip_graph_batch DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=1
