#!/usr/bin/env python3
"""Static VALU opcode mix of ONE kernel in libmppi_hip.so, from the gfx950 code objects themselves (llvm-objdump -d):
   tools/valu_static_mix.py 'k_rollout_pc<3, 3, 6, true, 0>' [out.json]
Used to price the `other` class of the VALU-issue floor (tools/summarize_profiles.py): the SQ counters give the launch's
vector instructions by class (add/mul/fma/trans f32, int32, int64, cvt) and lump the rest — moves, v_bitop3_b32, DPP forms,
lane swaps, selects, bit-field ops — into one number; this script says what that rest is made of. The producer waves of
k_rollout_pc are straight-line (the horizon groups are unrolled with compile-time slot indices), so the static mix of the
rest IS its dynamic mix up to the consumer's short loop body; the check printed at the end compares the static count with
the measured one."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def code_objects(so, workdir):
    """every gfx950 code object bundled in the library (one per translation unit)"""
    os.makedirs(workdir, exist_ok=True)
    fat = os.path.join(workdir, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for n, st in enumerate(starts):
        end = starts[n + 1] if n + 1 < len(starts) else len(blob)
        part = os.path.join(workdir, "bundle%d" % n)
        open(part, "wb").write(blob[st:end])
        co = os.path.join(workdir, "unit%d.co" % n)
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
            out.append(co)
    return out


def classify(op):
    """the SQ_INSTS_VALU_* class a vector opcode is counted in, or 'other:<kind>'"""
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "")
    dpp = base.endswith("_dpp")
    base = base.replace("_dpp", "")
    # packed fp32: the hardware counts one v_pk_add / _mul / _fma in SQ_INSTS_VALU_ADD / _MUL / _FMA_F32 like a plain one
    # (profiles/r05_valu_counter_classes.txt: tools/micro/valu_issue.hip under rocprofv3 --pmc) but issues it at 4.26 cycles, not 2.46:
    # tools/summarize_profiles.py prices each of the three classes by the static plain : packed share of the kernel's code object
    m = re.match(r"v_pk_(add|mul|fma)_f32$", base)
    if m:
        return "pk_%s_f32" % m.group(1)
    if base.startswith("v_pk_mov"):
        return "other:mov"
    if base.startswith("v_pk_"):
        return "other:misc"
    if re.match(r"v_(add|sub|subrev|min|max)_f32$", base):
        return "other:dpp_f32" if dpp else ("add_f32" if not base.startswith(("v_min", "v_max")) else "other:minmax")
    if re.match(r"v_mul(_legacy)?_f32$", base):
        return "mul_f32"
    if re.match(r"v_(fma|fmac|fmamk|fmaak|mad|mac)_f32$", base):
        return "fma_f32"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32$", base):
        return "trans_f32"
    if re.match(r"v_mad_[ui]64_[ui]32$", base) or base.endswith("_u64") or base.endswith("_i64") or base.endswith("_b64"):
        return "int64"
    if base.startswith("v_cvt_"):
        return "cvt"
    if base.startswith("v_mfma") or base.startswith("v_smfma"):
        return "mfma"
    if base in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
        return "other:dpp_mov" if dpp else "other:mov"
    if base.startswith("v_bitop3"):
        return "other:bitop3"
    if base.startswith("v_permlane"):
        return "other:permlane_swap"
    if base.startswith("v_cndmask"):
        return "other:cndmask"
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "other:lane"
    if base.startswith(("v_bfi", "v_bfe", "v_and_or", "v_lshl_or", "v_or3", "v_xad", "v_lshl_add", "v_add_lshl", "v_add3", "v_perm_b32", "v_alignbit")):
        return "other:bitfield3"
    if base.startswith("v_cmp"):
        return "other:cmp"
    if re.match(r"v_(add|sub|subrev|addc|subb)(_co)?(_ci)?_[ui]32$", base) or re.match(r"v_(and|or|xor|not|lshlrev|lshrrev|ashrrev)_b32$", base) \
            or re.match(r"v_(mul_lo|mul_hi|mul|min|max)_[ui](32|24)$", base) or re.match(r"v_(mul_u32_u24|mad_u32_u24|mad_i32_i24)$", base):
        return "int32"
    return "other:misc"


def main():
    pat = sys.argv[1]
    so = os.environ.get("MPPI_SO_PATH") or os.path.join(ROOT, "mppi-tf_amd", "libmppi_hip.so")
    hist, ops = collections.Counter(), collections.Counter()
    found = None
    for co in code_objects(so, os.path.join(ROOT, "build", "co")):
        dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--demangle", co], text=True, errors="replace")
        cur = None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                cur = m.group(1)
                continue
            if cur is None or pat not in cur or cur.endswith(".kd"):
                continue
            found = cur
            tok = line.strip().split()
            if not tok or not tok[0].startswith("v_"):
                continue
            ops[tok[0]] += 1
            hist[classify(tok[0])] += 1
        if found:
            break
    if not found:
        sys.exit("no kernel matching %r in %s" % (pat, so))
    out = {"kernel": found.split("(")[0], "classes": dict(sorted(hist.items())), "opcodes": dict(ops.most_common())}
    other = {k: v for k, v in hist.items() if k.startswith("other:")}
    out["other_total_static"] = sum(other.values())
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
