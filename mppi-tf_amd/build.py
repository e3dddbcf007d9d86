"""Build recipe for libmppi_hip.so (the product) — hipcc, gfx950 only, in-tree output.

The library is several translation units (csrc/*.hip): the C-ABI, and one object per (kernel family, action dimension)
— the rollout kernels are heavily unrolled templates and a single unit took 3-6 minutes. The units are compiled in
parallel (one hipcc -c per unit, a thread per CPU) into build/obj/ and linked into mppi-tf_amd/libmppi_hip.so.

-ffp-contract=off: the rollout arithmetic is the reference's op-by-op fp32 rounding (no fused
multiply-add), which is what makes sample costs bit-identical to an unfused fp32 evaluation on any IEEE CPU.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libmppi_hip.so")
# (source, defines, object stem). Heaviest first: the pool starts them first.
# The MLP units are compiled without the SLP vectoriser: what it finds in k_rollout_mlp_bx3's pieces becomes v_pk_*_f32, which
# serialise with the bf16 MFMA they are meant to hide behind (the other MLP kernels write their packed math explicitly). The
# 13-state unit too: the pairs it finds in the Fossen model cost two s_mov / v_mov each to assemble (0.30 -> 0.26 ms per step).
MLP_FLAGS = ["-fno-slp-vectorize"]
# ... and with builtin MFMAs in VGPR form: by default hipcc parks their accumulators in a0-a255 and pays a v_accvgpr_read / _write per
# register on either side of every layer of k_rollout_mlp32_bx3 / k_rollout_nnauv32_bx3 (the asm MFMAs of the other kernels name their registers themselves)
MLP_ONLY_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
UNITS = ([("mppi_launch_mlp.hip", ["MPPI_UNIT_A=%d" % a], "mlp_a%d" % a) for a in (3, 2, 1, 4)]
         + [("mppi_launch_pc.hip", ["MPPI_UNIT_A=%d" % a], "pc_a%d" % a) for a in (4, 3, 2, 1)]
         + [("mppi_launch_step.hip", ["MPPI_UNIT_A=%d" % a], "step_a%d" % a) for a in (4, 3, 2, 1)]
         + [("mppi_launch_tile.hip", ["MPPI_UNIT_A=%d" % a], "tile_a%d" % a) for a in (4, 3, 2, 1)]
         + [("mppi_launch_gen.hip", [], "gen"), ("mppi_learner.hip", [], "learner"), ("mppi_capi.hip", [], "capi")])
SOURCES = sorted({u[0] for u in UNITS})
HEADERS = ["mppi_device.hip.h", "mppi_ablate.hip.h", "mppi_kernels.hip.h", "mppi_mlp2.hip.h", "mppi_mlp_small.hip.h", "mppi_mlp32.hip.h",
           "mppi_handle.hip.h", "mppi_step.hip.h", "mppi_gen.hip.h", "mppi_mfma32.hip.h", "mppi_bx3.hip.h", "mppi_mlp32b.hip.h"]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def flags(extra=()):
    return ["-O3", "--offload-arch=" + ARCH, "-ffp-contract=off", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), *extra]


def _deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "mppi_c.h"), __file__]


def source_sha():
    """sha256 over the kernel sources: tags a measurement (profiles/*_latest.json) with the code it was taken on."""
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        hsh.update(open(os.path.join(CSRC, f), "rb").read())
    return hsh.hexdigest()[:16]


def stale(so=SO):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(d) > t for src in SOURCES for d in _deps(src))


def _jobs():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, int(os.environ.get("MPPI_BUILD_JOBS", "8"))))


def _compile_all(objdir, extra, force, verbose):
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def one(unit):
        src, defs, stem = unit
        obj = os.path.join(objdir, stem + ".o")
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in _deps(src)):
            return obj
        cmd = [cc, *flags(extra), *(MLP_FLAGS if stem.startswith("mlp_") or stem == "gen" else []), *(MLP_ONLY_FLAGS if stem.startswith("mlp_") or stem == "gen" else []), *["-D" + d for d in defs], "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(_jobs()) as pool:
        return list(pool.map(one, UNITS))


def _link(objs, out, verbose):
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    """Compile the HIP library if sources are newer than the in-tree .so. Returns its path."""
    if force or stale():
        objs = _compile_all(os.path.join(ROOT, "build", "obj", "default"), (), force, verbose)
        _link(objs, SO, verbose)
    return SO


def build_variant(name, defines=(), extra=(), force=False):
    """Variant builds of the SAME library (timing ablations, tools/ablate.py; the rocRAND-verbatim noise variant)
    -> build/variants/libmppi_hip_<name>.so"""
    d = os.path.join(ROOT, "build", "variants")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "libmppi_hip_%s.so" % name)
    if force or stale(out):
        objs = _compile_all(os.path.join(ROOT, "build", "obj", name), ["-D" + x for x in defines] + list(extra), force, False)
        _link(objs, out, False)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
