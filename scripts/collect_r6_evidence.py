#!/usr/bin/env python3
"""gpurun_out/r6_prof (scratch, written by scripts/gpu_r6_prof.sh on the GPU box) -> profiles/r06_* (tracked)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r6_prof"), os.path.join(ROOT, "profiles")


def line(fn):
    last = None
    for l in open(os.path.join(SRC, fn)):
        if l.startswith('{"metric"') or l.startswith('{"cnn"'):
            last = l
    return last


JSON = {"bench_driver_args.log": "r06_bench_driver_args.json", "bench_long.log": "r06_bench_long.json", "bench_b128.log": "r06_bench_b128.json",
        "bench_b128_dp.log": "r06_bench_b128_dp_world1.json", "bench_b512.log": "r06_bench_b512.json", "bench_b1024.log": "r06_bench_b1024.json",
        "bench_b4096.log": "r06_bench_b4096.json", "bench_b1024_rows10M.log": "r06_bench_b1024_rows10M.json",
        "bench_dp_native.log": "r06_bench_dp_native_world1.json", "bench_cnn.log": "r06_bench_cnn.json"}
COPY = {"bench_kernel_stats.csv": "r06_bench_kernel_stats.csv", "cnn_kernel_stats.csv": "r06_cnn_kernel_stats.csv",
        "dp_kernel_stats.csv": "r06_dp_kernel_stats.csv", "step_trace.txt": "r06_step_trace.txt", "pmc_summary_mlp.txt": "r06_pmc_summary_mlp.txt",
        "pmc_summary_cnn.txt": "r06_pmc_summary_cnn.txt", "pmc_traffic.json": "r06_pmc_traffic.json", "pmc_traffic_cnn.json": "r06_pmc_traffic_cnn.json"}
for a, b in JSON.items():
    try:
        open(os.path.join(DST, b), "w").write(json.dumps(json.loads(line(a)), indent=1) + "\n")
    except Exception as e:
        print("skip", a, e)
for a, b in COPY.items():
    if os.path.exists(os.path.join(SRC, a)):
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    else:
        print("missing", a)
print("done")
