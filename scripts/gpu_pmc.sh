#!/bin/bash
# PMC passes (own runs, kernel-trace only) for L2 hit rate / fetched bytes per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -c . $OUT/counters.txt
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-alt > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($pmc) rc=$?"
  python - "$OUT/pmc$i" <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv")
if not f:
    print("no counter csv in", d, glob.glob(d + "/*")); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "dsact" not in k: continue
    print(k, {c: round(v / cnt[(k, c)], 1) for c, v in acc[k].items()})
PY
done
