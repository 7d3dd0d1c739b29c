#!/bin/bash
# matrix-core utilisation per kernel (MLP and CNN workloads): --pmc MfmaUtil in its own pass, kernel-trace only
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
for wl in mlp cnn; do
  if [ $wl = mlp ]; then args="--steps 200 --warmup 20 --no-cpu-baseline --no-alt"; else args="--cnn-only --cnn-steps 40 --no-cpu-baseline"; fi
  timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/pmc_${wl}_mfma -o pmc -- python bench.py $args > $OUT/pmc_${wl}_mfma.log 2>&1
  echo "pmc $wl MfmaUtil rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
lines = ["MfmaUtil (rocprofv3 derived metric, % of cycles the matrix cores are busy, averaged over launches) per kernel;",
         "scripts/gpu_pmc3.sh: own --pmc pass with --kernel-trace only.", ""]
for wl in ("mlp", "cnn"):
    f = glob.glob("%s/pmc_%s_mfma/*counter_collection.csv" % (out, wl))
    if not f:
        lines.append("missing " + wl); continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != "MfmaUtil" or "dsact" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    lines.append("== %s workload" % wl)
    for k in sorted(acc, key=lambda k: -acc[k] / cnt[k]):
        lines.append("%-70s launches %5d  MfmaUtil %6.2f %%" % (k[:70], cnt[k], acc[k] / cnt[k]))
open(out + "/pmc_mfma_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
