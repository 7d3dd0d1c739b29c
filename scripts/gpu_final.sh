#!/bin/bash
# One gpurun call for the round's evidence: full GPU tests (+ parity report), smoke, the driver's bench command, a long
# bench, rocprofv3 kernel stats + launch-by-launch trace of the same bench, PMC passes (own runs, kernel-trace only),
# large batches, forced data-parallel legs. usage: gpurun --timeout 2400 -- 'bash scripts/gpu_final.sh [notests]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT; rm -f gpurun_out/parity_report.txt
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${1:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
  cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
fi
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "driver bench rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -1 $OUT/bench_driver.log | cut -c1-260
timeout 600 python bench.py --steps 4000 --warmup 400 > $OUT/bench.log 2>&1; echo "long bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-260
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1000 > $OUT/step_trace.txt && head -16 $OUT/step_trace.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$pmc -o pmc -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $OUT/pmc_$pmc.log 2>&1; echo "pmc $pmc rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
per = collections.defaultdict(dict)
for pmc in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
    f = glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, pmc), recursive=True)
    if not f:
        print("missing", pmc); continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    for k in acc:
        per[k][pmc] = acc[k] / cnt[k]; per[k]["launches"] = cnt[k]
lines = ["per kernel, averaged over its launches in `bench.py --steps 200 --warmup 20` (hipGraph replays; own --pmc pass per counter, --kernel-trace only);",
         "FETCH_SIZE / WRITE_SIZE in units of 1024 B, FETCH_SIZE doubled per the MI355X guide's gfx950 note; MfmaUtil = % of cycles the matrix cores are busy", ""]
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    fs, ws = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    lines.append("%-66s launches %5d  fetch %8.3f MB  write %8.3f MB  MfmaUtil %6.2f %%" % (k[:66], v["launches"], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, v.get("MfmaUtil", float("nan"))))
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
json.dump({"mlp": per}, open(out + "/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_MfmaUtil
for b in 512 1024 4096; do
  timeout 300 python bench.py --steps 1000 --warmup 100 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; tail -1 $OUT/bench_b$b.log | cut -c1-200
done
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/bench_dp_native.log 2>&1; echo "dp native rc=$?"; tail -1 $OUT/bench_dp_native.log | cut -c1-200
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt --dp-eager > $OUT/bench_dp_eager.log 2>&1; echo "dp eager rc=$?"; tail -1 $OUT/bench_dp_eager.log | cut -c1-200
# ---- CNN workload (configs[3]): bench object, kernel stats, PMC traffic per kernel (own passes, kernel-trace only)
timeout 400 python bench.py --cnn-only --cnn-steps 400 > $OUT/bench_cnn.log 2>&1; echo "cnn bench rc=$?"; grep '^{"cnn"' $OUT/bench_cnn.log | cut -c1-300
DSACT_NO_CHAIN_CNN=1 DSACT_NO_CONV_NARROW9=1 timeout 300 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline > $OUT/bench_cnn_r3_paths.log 2>&1; echo "cnn (tile-path trunks, LDS-tile layer 2) rc=$?"; grep '^{"cnn"' $OUT/bench_cnn_r3_paths.log | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cnn -o cnn -- python bench.py --cnn-only --cnn-steps 200 --no-cpu-baseline > $OUT/rocprof_cnn.log 2>&1; echo "rocprof cnn rc=$?"
cp $(find $OUT/prof_cnn -name "*kernel_stats.csv" | head -1) $OUT/cnn_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_cnn
for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmcc_$pmc -o pmc -- python bench.py --cnn-only --cnn-steps 40 --no-cpu-baseline > $OUT/pmcc_$pmc.log 2>&1; echo "pmc cnn $pmc rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
per = collections.defaultdict(dict)
for pmc in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
    f = glob.glob("%s/pmcc_%s/**/*counter_collection.csv" % (out, pmc), recursive=True)
    if not f:
        print("missing", pmc); continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    for k in acc:
        per[k][pmc] = acc[k] / cnt[k]; per[k]["launches"] = cnt[k]
n_upd = max([v["launches"] for k, v in per.items() if "k_gather_img" in k] + [1])
lines = ["cnn workload (configs[3], batch 256): per kernel, averaged over its launches in `bench.py --cnn-only --cnn-steps 40` (own --pmc pass per counter, --kernel-trace only);",
         "FETCH_SIZE / WRITE_SIZE in units of 1024 B, FETCH_SIZE doubled per the MI355X guide's gfx950 note; MfmaUtil = % of cycles the matrix cores are busy;",
         "per update = per launch x launches / updates (%d updates)" % n_upd, ""]
tot_f = tot_w = 0.0
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * kv[1]["launches"]):
    fs, ws = 2 * v.get("FETCH_SIZE", 0.0) * 1024 / 1e6, v.get("WRITE_SIZE", 0.0) * 1024 / 1e6
    per_upd = v["launches"] / n_upd
    tot_f += fs * per_upd; tot_w += ws * per_upd
    lines.append("%-62s launches %6d  fetch %8.3f MB  write %8.3f MB  per update %8.2f MB  MfmaUtil %6.2f %%" % (k[:62], v["launches"], fs, ws, (fs + ws) * per_upd, v.get("MfmaUtil", float("nan"))))
lines.append("")
lines.append("whole update: fetch %.1f MB + write %.1f MB = %.1f MB" % (tot_f, tot_w, tot_f + tot_w))
open(out + "/pmc_summary_cnn.txt", "w").write("\n".join(lines) + "\n")
json.dump({"cnn": per}, open(out + "/pmc_traffic_cnn.json", "w"), indent=1)
print("\n".join(lines[-3:]))
PY
rm -rf $OUT/pmcc_FETCH_SIZE $OUT/pmcc_WRITE_SIZE $OUT/pmcc_MfmaUtil
