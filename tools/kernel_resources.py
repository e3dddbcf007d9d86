#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels in libmppi_hip.so (from the gfx950 code object's metadata):
   tools/kernel_resources.py [name-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("MPPI_SO_PATH") or os.path.join(ROOT, "mppi-tf_amd", "libmppi_hip.so")
d = os.path.join(ROOT, "build", "co")
os.makedirs(d, exist_ok=True)
llvm = "/opt/rocm/lib/llvm/bin/"
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, os.path.join(d, "fatbin")])
subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + os.path.join(d, "fatbin"),
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(d, "lib.co")])
notes = subprocess.check_output([llvm + "llvm-readelf", "--notes", os.path.join(d, "lib.co")], text=True)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in notes.split("  - .agpr_count:")[1:]:
    blk = "  - .agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip() or g("name")
    if pat in name:
        print("%-70s vgpr %s (agpr %s) sgpr %s scratch %s B spills v/s %s/%s" % (
            name.split("(")[0][:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"),
            g("vgpr_spill_count"), g("sgpr_spill_count")))
