#!/usr/bin/env python3
"""A differently COMPILED library for A/B runs on one box: python scripts/build_variant.py <name> <hipcc flag> [...]
-> build/libdsact_<name>.so (objects in build/obj_<name>/), e.g.  python scripts/build_variant.py nt1 -DDSACT_NT_OPT=1
Use it with DSACT_LIB_PATH=$PWD/build/libdsact_<name>.so (scripts/gpu_r5_libs.sh: LIBS="nt1 nt2")."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

name, extra = sys.argv[1], sys.argv[2:]
obj = os.path.join(ROOT, "build", "obj_" + name)
os.makedirs(obj, exist_ok=True)
units = sorted(glob.glob(os.path.join(g.CSRC, "*.hip")))
jobs = [[g._hipcc()] + g.HIPCC_FLAGS + extra + ["-c", "-o", os.path.join(obj, os.path.splitext(os.path.basename(u))[0] + ".o"), u] for u in units]
with ThreadPoolExecutor(max_workers=8) as ex:
    for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            sys.exit(1)
out = os.path.join(ROOT, "build", "libdsact_%s.so" % name)
subprocess.run([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + sorted(glob.glob(os.path.join(obj, "*.o"))), check=True)
print("built", out)
