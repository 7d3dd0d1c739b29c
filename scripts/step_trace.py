#!/usr/bin/env python3
"""Launch-by-launch view of one update from a rocprofv3 --kernel-trace CSV.

usage: python scripts/step_trace.py gpurun_out/prof/bench_kernel_trace.csv [update_index] > profiles/rNN_step_trace.txt

An update starts at a `k_gather` dispatch; prints two consecutive updates (kernel, grid, duration, gap to the
previous launch's end) and the sums, so the per-launch numbers quoted in DESIGN.md can be re-derived.
"""
import csv
import sys


def main():
    path = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "dsact::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_gather(" in r[2]]
    if len(starts) < which + 3:
        which = max(0, len(starts) - 3)
    print("rocprofv3 --kernel-trace, %s: updates %d and %d of %d (an update = k_gather .. last launch before the next k_gather)"
          % (path.split("/")[-1], which, which + 1, len(starts)))
    for u in (which, which + 1):
        seg = rows[starts[u]:starts[u + 1]]
        print("update %d" % u)
        total, prev_end = 0.0, None
        for s, e, name, grid, wg in seg:
            gap = (s - prev_end) / 1000.0 if prev_end is not None else 0.0
            dur = (e - s) / 1000.0
            total += dur
            print("  %-62s blocks %5d x %3d  dur %6.2f us  gap %5.2f us" % (name[:62], grid // wg, wg, dur, gap))
            prev_end = e
        span = (seg[-1][1] - seg[0][0]) / 1000.0
        print("  sum of durations %.1f us; first start -> last end %.1f us; launches %d" % (total, span, len(seg)))


if __name__ == "__main__":
    main()
