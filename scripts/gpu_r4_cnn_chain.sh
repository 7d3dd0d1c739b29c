#!/bin/bash
# round 4: the CNN nets' twin trunks on the row-slice chains -- targeted parity tests, then the CNN bench leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_cnn_parity.py tests/test_hip_v1_cnn_parity.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/cnn_chain_tests.txt
cat gpurun_out/cnn_chain_tests.txt
