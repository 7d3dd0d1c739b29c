#!/bin/bash
# the whole GPU suite + smoke on the round's tree (what the driver runs at round end)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > gpurun_out/r6_suite.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " gpurun_out/r6_suite.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r6_smoke.log
