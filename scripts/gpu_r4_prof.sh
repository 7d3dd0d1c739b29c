#!/bin/bash
# round 4 evidence: driver-command + long bench lines, rocprofv3 kernel stats + a launch-by-launch trace of two consecutive
# updates of the bench (the two shapes of the pipelined graph), PMC passes (own runs, --kernel-trace only) for the MLP workload,
# forced data-parallel legs, batch legs. usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r4_prof.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4_prof; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
line() { grep '^{"metric"' $1 | tail -1; }
summ() { line $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
    for k in ('e2e', 'dsac_v1', 'alt', 'cnn'):
        if k in d: print('   %s %s' % (k, json.dumps({a: b for a, b in d[k].items() if a in ('value', 'ms_per_step', 'ms_per_iteration', 'sampler_ms_per_iteration')})))
except Exception as e:
    print('   parse error', e)
"; }
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.log 2>&1; echo "driver command rc=$?"; summ $OUT/bench_driver_args.log
timeout 600 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_long.log 2>&1; echo "long bench rc=$?"; summ $OUT/bench_long.log
DSACT_NO_PIPE=1 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_long_nopipe.log 2>&1; echo "long bench, DSACT_NO_PIPE=1 rc=$?"; summ $OUT/bench_long_nopipe.log
DSACT_NO_PIPE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_driver_nopipe.log 2>&1; echo "driver command, DSACT_NO_PIPE=1 rc=$?"; summ $OUT/bench_driver_nopipe.log
for b in 128; do
  timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
  DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b${b}_dp.log 2>&1; echo "batch $b dp rc=$?"; summ $OUT/bench_b${b}_dp.log
done
for b in 512 1024 4096; do
  timeout 300 python bench.py --steps 1000 --warmup 200 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; summ $OUT/bench_b$b.log
done
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_dp_native.log 2>&1; echo "dp native rc=$?"; summ $OUT/bench_dp_native.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1001 > $OUT/step_trace.txt && head -16 $OUT/step_trace.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_mlp_$pmc -o pmc -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt > $OUT/pmc_mlp_$pmc.log 2>&1; echo "pmc mlp $pmc rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
res = {}
for wl in ("mlp",):
    per = collections.defaultdict(dict)
    for pmc in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
        f = glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, wl, pmc), recursive=True)
        if not f:
            print("missing", wl, pmc); continue
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
            # the pipelined forward kernel comes in shapes: tell them apart by their grid
            name = r["Kernel_Name"]
            if "k_chain_fwdp" in name or "k_chain_bwd_pi<" in name:
                name += " grid=%s" % r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            acc[name] += float(r["Counter_Value"]); cnt[name] += 1
        for k in acc:
            per[k][pmc] = acc[k] / cnt[k]; per[k]["launches"] = cnt[k]
    res[wl] = per
    lines = ["%s workload: per kernel (pipelined forward / policy-backward launches per grid size = per shape), averaged over its launches in the bench (hipGraph replays; own --pmc pass per counter, --kernel-trace only);" % wl,
             "FETCH_SIZE / WRITE_SIZE in units of 1024 B, FETCH_SIZE doubled per the MI355X guide's gfx950 note; MfmaUtil = % of cycles the matrix cores are busy", ""]
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        fs, ws = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        lines.append("%-100s launches %5d  fetch %8.3f MB  write %8.3f MB  MfmaUtil %6.2f %%" % (k[:100], v["launches"], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, v.get("MfmaUtil", float("nan"))))
    open(out + "/pmc_summary_%s.txt" % wl, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_mlp_FETCH_SIZE $OUT/pmc_mlp_WRITE_SIZE $OUT/pmc_mlp_MfmaUtil 2>/dev/null; ls $OUT
