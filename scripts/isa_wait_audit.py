#!/usr/bin/env python3
"""Where does hipcc make a loop wait for EVERY outstanding load?  (build-container check, no GPU)

hipcc derives its `s_waitcnt vmcnt(N)` counts from the order of the requests it can see on a loop's entry and back edges.
A register-prefetch loop whose prologue requests its slots in another order than the body refills them, or whose body
refills / multiplies under a (wave-uniform) `if`, gets `vmcnt(0)` at the top of every trip: the look-ahead is lost and
every trip pays one L2 round trip (round 5: dw2_tile's streaming loop at batch >= 512 -- fixed: branch-free body, peeled
first trip; the throughput-regime first layer fat_gemm_x has the same shape and measured no gain from the fix).

usage: python scripts/isa_wait_audit.py [substring of a demangled kernel name ...]      (compiles the units of csrc/ to
assembly, ~40 s, cached in /tmp/dsact_isa/all.s while the sources are unchanged)
Prints, per innermost loop that contains vector-memory loads: its length, loads, MFMAs and its events in order
(M = MFMA, L = load, wN = s_waitcnt vmcnt(N), B = barrier, d / r = LDS write / read; a count follows a repeated event) -- a `w0`
in front of the MFMAs of a loop that issues its loads AFTER them is the pattern to look for."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dsac-v2_amd", "csrc")
OUT = "/tmp/dsact_isa"


def build():
    os.makedirs(OUT, exist_ok=True)
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    tag, asm = os.path.join(OUT, "tag"), os.path.join(OUT, "all.s")
    if os.path.exists(asm) and os.path.exists(tag) and open(tag).read() == h.hexdigest():
        return asm
    # every translation unit of the library (the host side + one unit per kernel family, csrc/dsact_tu.h), side by side
    from concurrent.futures import ThreadPoolExecutor

    units = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

    def one(u):
        out = os.path.join(OUT, u + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                        "-Wno-cuda-compat", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                        os.path.join(CSRC, u)],
                       check=True, stderr=subprocess.DEVNULL)
        return out

    with ThreadPoolExecutor(max_workers=8) as ex:
        parts = list(ex.map(one, units))
    with open(asm, "w") as f:
        for part in parts:
            f.write(open(part).read())
    open(tag, "w").write(h.hexdigest())
    return asm


def main():
    pats = sys.argv[1:]
    lines = open(build()).read().split("\n")
    names = [re.match(r"^(_ZN5dsact\w+):", l).group(1) for l in lines if re.match(r"^_ZN5dsact\w+:", l)]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    for name, dm in zip(names, dem):
        if pats and not any(p in dm for p in pats):
            continue
        a = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
        lab = {}
        for i in range(a, b):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                lab[m.group(1)] = i
        rows = []
        for i in range(a, b):
            m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
            if not m or m.group(1) not in lab or lab[m.group(1)] > i:
                continue
            seg = lines[lab[m.group(1)]:i + 1]
            if any("Loop Header" in s for s in seg[1:]):
                continue   # not innermost
            ev = []
            for s in seg:
                if "v_mfma" in s:
                    ev.append("M")
                elif re.search(r"\s(global_load|buffer_load)", s):
                    ev.append("L")
                elif "vmcnt" in s:
                    ev.append("w" + re.search(r"vmcnt\((\d+)\)", s).group(1))
                elif "s_barrier" in s:
                    ev.append("B")
                elif re.search(r"\sds_write|\sds_store", s):
                    ev.append("d")
                elif re.search(r"\sds_read|\sds_load", s):
                    ev.append("r")
            if "L" not in ev or (ev.count("L") < 2 and "M" not in ev):
                continue
            out = []
            for e in ev:
                if out and out[-1][0] == e:
                    out[-1][1] += 1
                else:
                    out.append([e, 1])
            rows.append("    loop %-12s %4d lines  loads %3d  mfma %4d  %-11s %s" % (
                m.group(1), len(seg), ev.count("L"), ev.count("M"), "DRAINS (w0)" if "w0" in ev and "M" in ev else "",
                " ".join(("%s%d" % (e, n) if n > 1 else e) for e, n in out)[:300]))
        if rows:
            print(dm[:110])
            print("\n".join(rows))


if __name__ == "__main__":
    main()
