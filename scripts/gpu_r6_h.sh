#!/bin/bash
# round 6: what bounds the throughput-regime forward at batch 1024 -- PMC (own passes, --kernel-trace only): matrix-core busy, active
# cycles (-> the clock the kernels actually ran at), per kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/h; rm -rf $OUT; mkdir -p $OUT
for pmc in MfmaUtil GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU; do
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p_$pmc -o pmc -- python bench.py --steps 100 --warmup 20 --batch 1024 --headline-only > $OUT/p_$pmc.log 2>&1; echo "pmc $pmc rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
per = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for d in glob.glob(out + "/p_*/"):
    pmc = d.rstrip("/").split("p_")[-1]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print("missing", pmc); continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != pmc or "dsact" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    for k in acc: per[k][pmc] = acc[k] / cnt[k]; per[k]["n"] = cnt[k]
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if t and pmc == "GRBM_GUI_ACTIVE":
        for r in csv.DictReader(open(t[0])):
            if "dsact" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = []
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    du = sum(dur[k]) / max(1, len(dur[k])) / 1e3
    lines.append("%-80s n %4d dur %7.2f us %s" % (k[:80], v["n"], du, "  ".join("%s=%.4g" % (a, b) for a, b in sorted(v.items()) if a != "n")))
open(out + "/pmc_b1024.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/p_*/
