#!/usr/bin/env python3
"""Launch-by-launch view of one update from a rocprofv3 --kernel-trace CSV.

usage: python scripts/step_trace.py gpurun_out/prof/bench_kernel_trace.csv [update_index] > profiles/rNN_step_trace.txt

Prints two consecutive updates (kernel, grid, duration, gap to the previous launch's end) and the sums, so the
per-launch numbers quoted in DESIGN.md can be re-derived.
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
    # an update ends with the launch that closes it (k_stage_table, or k_adam on the unfused path); a k_gather in front
    # of it belongs to it (graph replays with the merged gather have one k_gather per graph, not per update)
    updates, cur = [], []
    for r in rows:
        cur.append(r)
        if "k_stage_table(" in r[2] or "k_adam(" in r[2]:
            updates.append(cur)
            cur = []
    if len(updates) < which + 2:
        which = max(0, len(updates) - 2)
    n_gather = sum(1 for r in rows if "k_gather(" in r[2])
    print("rocprofv3 --kernel-trace, %s: updates %d and %d of %d (%d k_gather launches in the trace)"
          % (path.split("/")[-1], which, which + 1, len(updates), n_gather))
    for u in (which, which + 1):
        seg = updates[u]
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
