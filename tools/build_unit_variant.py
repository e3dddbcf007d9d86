#!/usr/bin/env python3
"""A variant of the library that differs from the product in ONE translation unit (a study build in half a minute instead of a full
variant's two and a half):  tools/build_unit_variant.py <name> <unit stem: gen | pc_a3 | step_a2 | ...> [DEFINE ...]
-> build/variants/libmppi_hip_<name>.so = build/obj/default/*.o with <stem>.o recompiled under -D<DEFINE>. Run with MPPI_SO_PATH=<that file>."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mppi_tf_amd.build as b

name, stem, defines = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()  # the other units' objects
unit = next(u for u in b.UNITS if u[2] == stem)
objdir = os.path.join(b.ROOT, "build", "obj", name)
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, stem + ".o")
mlp = stem.startswith("mlp_") or stem == "gen"
cmd = [b.hipcc(), *b.flags(["-D" + d for d in defines]), *(b.MLP_FLAGS + b.MLP_ONLY_FLAGS if mlp else []), *["-D" + d for d in unit[1]], "-c",
       os.path.join(b.CSRC, unit[0]), "-o", obj]
subprocess.check_call(cmd)
objs = [obj if u[2] == stem else os.path.join(b.ROOT, "build", "obj", "default", u[2] + ".o") for u in b.UNITS]
out = os.path.join(b.ROOT, "build", "variants", "libmppi_hip_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
b._link(objs, out, False)
print(out)
