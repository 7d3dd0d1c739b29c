#!/usr/bin/env python3
"""Launch-by-launch view of one update from a rocprofv3 --kernel-trace CSV.

usage: python scripts/step_trace.py gpurun_out/prof/bench_kernel_trace.csv [update_index] [--bursts] > profiles/rNN_step_trace.txt

Prints two consecutive updates (kernel, grid, duration, gap to the previous launch's end) and the sums, so the
per-launch numbers quoted in DESIGN.md can be re-derived. --bursts: additionally one line per burst of launches
(bursts are separated by > 300 us of idle GPU: the timed regions of a short bench run) with its span, the sum of its
kernel durations and its idle time -- where a short region's time goes.
"""
import csv
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    close_on = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--close=")]   # e.g. --close=k_conv_dw_reduce (CNN)
    path = args[0]
    which = int(args[1]) if len(args) > 1 else 1000
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "dsact::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
    rows.sort()
    # an update ends with the launch that closes it (k_stage_table, or k_adam on the unfused path); a k_gather in front
    # of it belongs to it (graph replays with the merged gather have one k_gather per graph, not per update)
    # (row-slice chain path: k_dw2 closes it unless the unfused optimiser pass k_adam / the split-K sum follows)
    updates, cur = [], []
    for i, r in enumerate(rows):
        cur.append(r)
        nxt = rows[i + 1][2] if i + 1 < len(rows) else ""
        closes = "k_stage_table(" in r[2] or "k_adam(" in r[2] or \
            (("k_dw2<" in r[2] or "k_chain_bwd_pi<" in r[2] or "k_chain_bwd_qt<" in r[2] or "k_chain_bwd_qpt<" in r[2])
             and "k_adam(" not in nxt and "k_sum_parts(" not in nxt and "k_dw2<" not in nxt and "AllReduce" not in nxt)
        if close_on:
            closes = any(c in r[2] for c in close_on)
        if closes:
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
    if "--bursts" in sys.argv:
        bursts(rows)


def bursts(rows, idle_us=300.0):
    out, cur = [], [rows[0]]
    for r in rows[1:]:
        if (r[0] - cur[-1][1]) / 1000.0 > idle_us:
            out.append(cur)
            cur = []
        cur.append(r)
    out.append(cur)
    print("bursts of launches (separated by > %.0f us idle): %d" % (idle_us, len(out)))
    for b in out[-12:]:
        span = (b[-1][1] - b[0][0]) / 1000.0
        busy = sum(e - s for s, e, *_ in b) / 1000.0
        closers = sum(1 for r in b if "k_dw2<" in r[2] or "k_stage_table(" in r[2])
        first = ", ".join("%s %.1f" % (r[2].split("::")[-1].split("(")[0][:14], (r[1] - r[0]) / 1000.0) for r in b[:7])
        print("  %4d launches (%3d closing)  span %9.1f us  busy %9.1f us  idle %7.1f us | first: %s"
              % (len(b), closers, span, busy, span - busy, first))


if __name__ == "__main__":
    main()
