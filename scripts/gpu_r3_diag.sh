#!/bin/bash
# GPU diagnostic call: scripts/diag_case.py for the activation pairs given as "va:pa" arguments
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3_diag
for pair in "$@"; do
  va=${pair%%:*}; pa=${pair##*:}
  timeout 300 python scripts/diag_case.py $va $pa > gpurun_out/r3_diag/diag_${va}_${pa}.log 2>&1
  echo "== $pair rc=$?"; tail -40 gpurun_out/r3_diag/diag_${va}_${pa}.log
done
