#!/bin/bash
# round 3 evidence: rocprofv3 kernel stats + one-update trace of the bench, PMC passes (own runs, --kernel-trace only) for
# the MLP and the CNN workloads, forced data-parallel legs, the batch-128 leg. usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r3_prof.sh ["<pytest -k>"]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_prof; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -n "${1:-}" ]; then
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$1" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_gpu.log | tail -25
fi
summ() { tail -1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
    if 'e2e' in d: print('   e2e', json.dumps({k: v for k, v in d['e2e'].items() if k != 'note'})[:600])
except Exception as e:
    print('   parse error', e)
"; }
timeout 600 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $OUT/bench_long.log 2>&1; echo "long bench rc=$?"; summ $OUT/bench_long.log
for b in 128; do
  timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
  DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b${b}_dp.log 2>&1; echo "batch $b dp rc=$?"; summ $OUT/bench_b${b}_dp.log
done
for b in 512 2048; do
  timeout 300 python bench.py --steps 2000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
done
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_dp_native.log 2>&1; echo "dp native rc=$?"; summ $OUT/bench_dp_native.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1000 > $OUT/step_trace.txt && head -14 $OUT/step_trace.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
for wl in mlp cnn; do
  if [ $wl = mlp ]; then CMD="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt"; else CMD="python bench.py --cnn-only --cnn-steps 60 --no-cpu-baseline"; fi
  for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$pmc -o pmc -- $CMD > $OUT/pmc_${wl}_$pmc.log 2>&1; echo "pmc $wl $pmc rc=$?"
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cnn -o cnn -- python bench.py --cnn-only --cnn-steps 100 --no-cpu-baseline > $OUT/rocprof_cnn.log 2>&1; echo "rocprof cnn rc=$?"
cp $(find $OUT/prof_cnn -name "*kernel_stats.csv" | head -1) $OUT/cnn_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_cnn
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
res = {}
for wl in ("mlp", "cnn"):
    per = collections.defaultdict(dict)
    for pmc in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
        f = glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, wl, pmc), recursive=True)
        if not f:
            print("missing", wl, pmc); continue
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
            acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
        for k in acc:
            per[k][pmc] = acc[k] / cnt[k]; per[k]["launches"] = cnt[k]
    res[wl] = per
    lines = ["%s workload: per kernel, averaged over its launches in the bench (hipGraph replays for the MLP nets; own --pmc pass per counter, --kernel-trace only);" % wl,
             "FETCH_SIZE / WRITE_SIZE in units of 1024 B, FETCH_SIZE doubled per the MI355X guide's gfx950 note; MfmaUtil = % of cycles the matrix cores are busy", ""]
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        fs, ws = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        lines.append("%-72s launches %5d  fetch %8.3f MB  write %8.3f MB  MfmaUtil %6.2f %%" % (k[:72], v["launches"], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, v.get("MfmaUtil", float("nan"))))
    open(out + "/pmc_summary_%s.txt" % wl, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_mlp_* $OUT/pmc_cnn_*  2>/dev/null; ls $OUT
