#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels in libmppi_hip.so (from the metadata of its gfx950 code objects, one per
translation unit):   tools/kernel_resources.py [name-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from valu_static_mix import LLVM, code_objects  # noqa: E402

so = os.environ.get("MPPI_SO_PATH") or os.path.join(ROOT, "mppi-tf_amd", "libmppi_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for co in code_objects(so, os.path.join(ROOT, "build", "co")):
    notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
    for blk in notes.split("  - .agpr_count:")[1:]:
        blk = "  - .agpr_count:" + blk
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip() or g("name")
        if pat in name:
            print("%-70s vgpr %s (agpr %s) sgpr %s lds %s B scratch %s B spills v/s %s/%s" % (
                name.split("(")[0][:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
