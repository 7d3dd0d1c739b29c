#!/bin/bash
# round 6 evidence from ONE tree: the driver's command (full line incl. cpu_baseline port, e2e, e2e_si8, cnn) + a long run, rocprofv3
# kernel stats + a launch-by-launch trace of two consecutive updates, PMC passes (own runs, --kernel-trace only) for the MLP and
# the CNN workload, batch legs, forced data-parallel legs. usage: gpurun --timeout 3000 -- 'bash scripts/gpu_r6_prof.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_prof; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
line() { grep '^{"metric"' $1 | tail -1; }
summ() { line $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
    for k in ('fast', 'e2e', 'e2e_si8', 'dsac_v1', 'alt', 'cnn', 'cpu_baseline'):
        if k in d: print('   %s %s' % (k, json.dumps({a: b for a, b in d[k].items() if a in ('value', 'ms_per_step', 'ms_per_iteration', 'sampler_ms_per_iteration', 'update_us_through_the_surface', 'ungrouped', 'kind')})[:400]))
except Exception as e:
    print('   parse error', e)
"; }
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.log 2>&1; echo "driver command rc=$?"; summ $OUT/bench_driver_args.log
timeout 600 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_long.log 2>&1; echo "long bench rc=$?"; summ $OUT/bench_long.log
for b in 128; do
  timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
  DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b${b}_dp.log 2>&1; echo "batch $b dp rc=$?"; summ $OUT/bench_b${b}_dp.log
done
for b in 512 1024 4096; do
  timeout 300 python bench.py --steps 1000 --warmup 200 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
done
timeout 600 python bench.py --steps 1000 --warmup 200 --batch 1024 --rows 10000000 --no-cpu-baseline --no-alt > $OUT/bench_b1024_rows10M.log 2>&1; echo "batch 1024, 10M-row ring (configs[4], one GPU's leg) rc=$?"; summ $OUT/bench_b1024_rows10M.log
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_dp_native.log 2>&1; echo "dp native rc=$?"; summ $OUT/bench_dp_native.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --headline-only > $OUT/rocprof.log 2>&1; echo "rocprof (headline configuration alone) rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1001 > $OUT/step_trace.txt && head -16 $OUT/step_trace.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
# the data-parallel graph with the collective forced at world 1: what RCCL's one-rank kernel and k_adam_pack cost per update
DSACT_BENCH_FORCE_DP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dp -o dp -- python bench.py --steps 2000 --warmup 200 --headline-only > $OUT/rocprof_dp.log 2>&1; echo "rocprof dp rc=$?"
cp $(find $OUT/prof_dp -name "*kernel_stats.csv" | head -1) $OUT/dp_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_dp
for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_mlp_$pmc -o pmc -- python bench.py --steps 200 --warmup 20 --headline-only > $OUT/pmc_mlp_$pmc.log 2>&1; echo "pmc mlp $pmc rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
res = {}
per = collections.defaultdict(dict)
for pmc in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
    f = glob.glob("%s/pmc_mlp_%s/**/*counter_collection.csv" % (out, pmc), recursive=True)
    if not f:
        print("missing mlp", pmc); continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    for k in acc:
        per[k][pmc] = acc[k] / cnt[k]; per[k]["launches"] = cnt[k]
res["mlp"] = per
lines = ["mlp workload (Humanoid 3x256, batch 256): per kernel, averaged over its launches in `bench.py --steps 200 --warmup 20` (hipGraph replays; own --pmc pass per counter, --kernel-trace only);",
         "FETCH_SIZE / WRITE_SIZE in units of 1024 B, FETCH_SIZE doubled per the MI355X guide's gfx950 note; MfmaUtil = % of cycles the matrix cores are busy", ""]
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    fs, ws = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    lines.append("%-100s launches %5d  fetch %8.3f MB  write %8.3f MB  MfmaUtil %6.2f %%" % (k[:100], v["launches"], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, v.get("MfmaUtil", float("nan"))))
# bytes per update of the timed graph's launches (replays only: launches per update from the kernel counts)
open(out + "/pmc_summary_mlp.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_mlp_FETCH_SIZE $OUT/pmc_mlp_WRITE_SIZE $OUT/pmc_mlp_MfmaUtil 2>/dev/null
# ---- CNN workload (configs[3]): bench object, kernel stats, PMC traffic per kernel -- the SAME tree
timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline > $OUT/bench_cnn.log 2>&1; echo "cnn bench rc=$?"; grep '^{"cnn"' $OUT/bench_cnn.log | cut -c1-300
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
ls $OUT
