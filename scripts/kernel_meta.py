#!/usr/bin/env python3
"""Register / scratch / kernarg usage of the gfx950 kernels in dsac-v2_amd/lib/libdsact.so (build-container check, no GPU):
python scripts/kernel_meta.py [substring ...]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
so = "dsac-v2_amd/lib/libdsact.so"
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, d + "/fat.bin"], check=True)
    # one offload bundle per translation unit (csrc/dsact_api.hip + the kernel-family units), concatenated in the section
    blob = open(d + "/fat.bin", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    notes = ""
    for i, st in enumerate(starts):
        en = starts[i + 1] if i + 1 < len(starts) else len(blob)
        open(d + "/b%d.bin" % i, "wb").write(blob[st:en])
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + d + "/b%d.bin" % i,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + d + "/k%d.co" % i], capture_output=True)
        if r.returncode != 0:
            continue
        notes += subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/k%d.co" % i], capture_output=True, text=True).stdout
for k in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    if len(sys.argv) > 1 and not any(s in name for s in sys.argv[1:]):
        continue
    g = lambda f: (re.search(r"\.%s:\s+(\d+)" % f, k) or [None, "?"])[1]
    print("%-72s vgpr %3s spill %s scratch %s kernarg %s" % (name[:72], g("vgpr_count"), g("vgpr_spill_count"),
                                                              g("private_segment_fixed_size"), g("kernarg_segment_size")))
