#!/bin/bash
# kernel durations UNDER the --pmc MfmaUtil pass (same bench, same graph replays) beside the MfmaUtil values: is the
# counter taken at the bench's launch cadence?  usage: gpurun -- 'bash scripts/gpu_pmc_cadence.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_cadence; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || exit 1
timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/p -o pmc -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $OUT/run.log 2>&1; echo "rc=$?"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
dur, cnt = collections.defaultdict(float), collections.Counter()
f = glob.glob(out + "/p/**/*kernel_trace.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "dsact" in k:
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0; cnt[k] += 1
util, ucnt = collections.defaultdict(float), collections.Counter()
f = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "MfmaUtil" and "dsact" in r["Kernel_Name"]:
        util[r["Kernel_Name"]] += float(r["Counter_Value"]); ucnt[r["Kernel_Name"]] += 1
lines = ["rocprofv3 --pmc MfmaUtil --kernel-trace -- python bench.py --steps 200 --warmup 20: per kernel, average duration UNDER the counter pass and MfmaUtil"]
for k in sorted(dur, key=lambda k: -dur[k]):
    lines.append("%-62s launches %5d  avg %7.2f us under --pmc   MfmaUtil %6.2f %%" % (k[:62], cnt[k], dur[k] / cnt[k], util[k] / max(1, ucnt[k])))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
rm -rf $OUT/p
