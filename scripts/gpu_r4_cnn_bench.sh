#!/bin/bash
# round 4: CNN leg (configs[3]): twin trunks on the chains (parallel trunk workgroups / both trunks in one workgroup / 4-row
# slices) vs on the stage tiles (DSACT_NO_CHAIN_CNN=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" = test ]; then
  timeout 900 python -m pytest tests/test_hip_cnn_parity.py tests/test_hip_v1_cnn_parity.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/cnn_chain_tests.txt
fi
for v in ${VARIANTS:-par}; do
  unset DSACT_NO_CHAIN_CNN DSACT_TWIN_SEQ DSACT_CHAIN_RG DSACT_TWIN_NO_MERGE DSACT_NO_CONV_NARROW9 DSACT_NO_CONV_FWD64 DSACT_NO_CONV_FWD32X64 DSACT_DFEAT64
  case $v in
    tile) export DSACT_NO_CHAIN_CNN=1;;
    seq) export DSACT_TWIN_SEQ=1;;
    par_rg1) export DSACT_CHAIN_RG=1;;
    nomerge) export DSACT_TWIN_NO_MERGE=1;;
    nonarrow9) export DSACT_NO_CONV_NARROW9=1;;
    nofwd64) export DSACT_NO_CONV_FWD64=1;;
    no32x64) export DSACT_NO_CONV_FWD32X64=1;;
    dfeat64) export DSACT_DFEAT64=1;;
  esac
  timeout 300 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline 2>/dev/null | grep '^{"cnn"' > gpurun_out/r04_cnn_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04_cnn_$v.json"))["cnn"]
print("$v", round(d["value"],1), "steps/s", round(d["ms_per_step"]*1000,1), "us", len(d.get("kernels",[])), "launches")
for k in d.get("kernels",[]): print("   %-18s %8.2f us %6d blocks" % (k["name"], k["us"], k["blocks"]))
PY
done 2>&1 | tee gpurun_out/r04_cnn_ab.txt
