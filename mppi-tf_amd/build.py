"""Build recipe for libmppi_hip.so (the product) — hipcc, gfx950 only, in-tree output.

-ffp-contract=off: the rollout arithmetic is the reference's op-by-op fp32 rounding (no fused
multiply-add), which is what makes sample costs bit-identical to an unfused fp32 evaluation on any IEEE CPU.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libmppi_hip.so")
SOURCES = ["mppi_capi.hip"]
HEADERS = ["mppi_device.hip.h", "mppi_kernels.hip.h", "mppi_mlp2.hip.h", "mppi_mlp_small.hip.h", "mppi_mlp32.hip.h"]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def command(out=SO, extra=()):
    return [hipcc(), "-O3", "--offload-arch=" + ARCH, "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
            "-I", os.path.join(ROOT, "include"), *extra,
            *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]


def source_sha():
    """sha256 over the kernel sources: tags a measurement (profiles/*_latest.json) with the code it was taken on."""
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        hsh.update(open(os.path.join(CSRC, f), "rb").read())
    return hsh.hexdigest()[:16]


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "mppi_c.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP library if sources are newer than the in-tree .so. Returns its path."""
    if force or stale():
        cmd = command()
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


def build_variant(name, defines=(), extra=()):
    """Timing-only variant builds for ablation (tools/ablate.py) -> build/variants/libmppi_hip_<name>.so"""
    d = os.path.join(ROOT, "build", "variants")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "libmppi_hip_%s.so" % name)
    subprocess.check_call(command(out, ["-D" + x for x in defines] + list(extra)))
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
