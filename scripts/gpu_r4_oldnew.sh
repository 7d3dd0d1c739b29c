#!/bin/bash
# A/B of two builds of the library on one box (DSACT_LIB_PATH): batch 1024 and the headline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
summ() { grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'steps/s', round(1000*d['ms_per_step'],2), 'us', [ (k['name'],k['us']) for k in (d.get('kernels') or []) ] if isinstance(d.get('kernels'), list) else '')"; }
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export DSACT_LIB_PATH=$PWD/dsac-v2_amd/lib/libdsact_old.so; else unset DSACT_LIB_PATH; fi
  echo "$v b1024"; timeout 300 python bench.py --steps 1000 --warmup 100 --batch 1024 --no-cpu-baseline --no-alt 2>/dev/null | summ
  echo "$v long"; timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt 2>/dev/null | summ
done; done
