#!/bin/bash
# quick legs: the driver's command, a 4,000-step run, batch 1024, CNN
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
summ() { grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'steps/s', round(1000*d['ms_per_step'],2), 'us', [ (k['name'],k['us']) for k in (d.get('kernels') or []) ] if isinstance(d.get('kernels'), list) else '')"; }
echo driver; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | summ
echo long; timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt 2>/dev/null | summ
echo b1024; timeout 300 python bench.py --steps 1000 --warmup 100 --batch 1024 --no-cpu-baseline --no-alt 2>/dev/null | summ
echo b512; timeout 300 python bench.py --steps 1000 --warmup 100 --batch 512 --no-cpu-baseline --no-alt 2>/dev/null | summ
