#!/usr/bin/env python3
"""Summarise gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the committed evidence:
   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of bench.py
   profiles/<tag>_pmc_summary.json     per-kernel averages of the PMC counters (separate passes)
   profiles/traffic_latest.json        HBM bytes per launch of the dominant kernel      } read by bench.py, which uses them
   profiles/valu_latest.json           its vector instructions per launch by class      } only when `code_sha` matches the
                                       x the issue cycles per instruction of the class  } kernel sources it runs
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads (x2 correction); WRITE_SIZE is exact for dword stores.
Issue cycles come from profiles/<micro>_valu_issue.json (tools/micro/valu_issue.hip run on the same part), rows with 4
waves per SIMD — what k_rollout_pc runs at."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
micro = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
import mppi_tf_amd  # noqa: E402
code_sha = mppi_tf_amd.build.source_sha()

stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))

summary = collections.defaultdict(dict)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if "mppi::" not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in acc.items():
        for c, v in cs.items():
            summary[name][c] = sum(v) / len(v)
            summary[name]["launches_" + c] = len(v)

dominant = next((k for k in summary if "k_rollout" in k), None)
out = {"tag": tag, "code_sha": code_sha, "kernels": summary}
kname = dominant.replace("void ", "") if dominant else None
if dominant and "FETCH_SIZE" in summary[dominant] and "WRITE_SIZE" in summary[dominant]:
    fetch_kib, write_kib = summary[dominant]["FETCH_SIZE"], summary[dominant]["WRITE_SIZE"]
    hbm = (2.0 * fetch_kib + write_kib) * 1024.0
    traffic = {"kernel": kname, "code_sha": code_sha, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
               "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
               "hbm_bytes_per_launch": hbm, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tag " + tag}
    out["traffic"] = traffic
    if "k_rollout_pc" in dominant:  # bench.py's default workload reads this file
        json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)

mfile = os.path.join(dst, micro + "_valu_issue.json")
if dominant and "k_rollout_pc" in dominant and "SQ_INSTS_VALU_MUL_F32" in summary[dominant] and os.path.exists(mfile):
    rows = [r for r in json.load(open(mfile)) if r["waves_per_simd_median"] == 4]
    cyc = {r["op"]: r["cyc_per_inst_per_simd"] for r in rows}
    k = summary[dominant]
    classes = {"add_f32": k["SQ_INSTS_VALU_ADD_F32"], "mul_f32": k["SQ_INSTS_VALU_MUL_F32"], "fma_f32": k["SQ_INSTS_VALU_FMA_F32"],
               "trans_f32": k["SQ_INSTS_VALU_TRANS_F32"], "int32": k["SQ_INSTS_VALU_INT32"], "int64": k["SQ_INSTS_VALU_INT64"],
               "cvt": k["SQ_INSTS_VALU_CVT"]}
    classes["other"] = k["SQ_INSTS_VALU"] - sum(classes.values())
    # `other` priced from what it is made of: the static opcode mix of the kernel's code object (tools/valu_static_mix.py;
    # the producer waves are straight-line code, so static shares are dynamic shares), each kind at its measured issue cost
    import subprocess
    mixfile = os.path.join(dst, tag + "_valu_static_mix.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_static_mix.py"), kname.replace("void ", "").replace("mppi::", ""), mixfile],
                          stdout=subprocess.DEVNULL)
    mix = {c: n for c, n in json.load(open(mixfile))["classes"].items() if c.startswith("other:")}
    kind_price = {"other:mov": cyc["v_mov_b32"], "other:bitop3": cyc["v_bitop3_b32"], "other:dpp_f32": cyc["v_add_f32_dpp quad_perm"],
                  "other:dpp_mov": cyc["v_mov_b32_dpp row_mirror"], "other:permlane_swap": cyc["v_mov_b32_dpp row_mirror"],
                  "other:cndmask": cyc["v_cndmask_b32_e64 (mask in s[44:45])"], "other:lane": cyc["v_readlane_b32"],
                  "other:minmax": cyc["v_max_f32"], "other:cmp": cyc["v_xor_b32"], "other:bitfield3": cyc["v_bitop3_b32"],
                  "other:misc": cyc["v_bitop3_b32"]}
    other_price = sum(n * kind_price[c] for c, n in mix.items()) / max(1, sum(mix.values()))
    price = {"add_f32": cyc["v_add_f32"], "mul_f32": cyc["v_mul_f32"], "fma_f32": cyc["v_fma_f32"],
             "trans_f32": (cyc["v_log_f32"] + cyc["v_sqrt_f32"] + cyc["v_sin_f32"]) / 3, "int32": cyc["v_xor_b32"],
             "int64": cyc["v_mad_u64_u32 (+0)"], "cvt": cyc["v_cvt_f32_u32"], "other": other_price}
    valu = {"kernel": kname, "code_sha": code_sha, "tag": tag, "simds": 1024, "clock_mhz": 2400.0,
            "other_static_mix": mix, "other_kind_cycles": kind_price,
            "insts_per_launch": classes, "cycles_per_inst": price,
            "active_quad_cycles_per_launch": k.get("SQ_ACTIVE_INST_VALU"),
            "source": "instructions: rocprofv3 --pmc SQ_INSTS_VALU* (tag %s, own pass); cycles: profiles/%s_valu_issue.json at 4 waves "
                      "per SIMD (ISA of the timed loops: profiles/%s_valu_issue_isa.txt)" % (tag, micro, micro),
            "classes": "the SQ counters' classes: add/mul/fma/trans f32, int32, int64 (v_mad_u64_u32: the Philox products), cvt; "
                       "`other` = SQ_INSTS_VALU minus those (v_bitop3_b32 of the Philox rounds, moves, DPP and lane-swap forms), "
                       "priced by its static opcode mix (other_static_mix x other_kind_cycles; lane swaps at the DPP rate). Packed f32 instructions count in add/mul at the unpacked price although "
                       "they issue at ~4.2 cycles: the floor is a lower bound"}
    out["valu"] = valu
    json.dump(valu, open(os.path.join(dst, "valu_latest.json"), "w"), indent=1)
if dominant and "k_rollout_mlp" in dominant and "SQ_VALU_MFMA_BUSY_CYCLES" in summary[dominant] and "GRBM_GUI_ACTIVE" in summary[dominant]:
    k = summary[dominant]  # the MLP path: how busy the matrix pipe was, and what shared the vector pipe with it
    mfma = {"kernel": kname, "code_sha": code_sha,
            "mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (k["GRBM_GUI_ACTIVE"] / 8.0),
            "mfma_insts_per_launch": k.get("SQ_INSTS_MFMA"),
            "other_vector_insts_per_launch": (k["SQ_INSTS_VALU"] - k["SQ_INSTS_MFMA"]) if "SQ_INSTS_VALU" in k and "SQ_INSTS_MFMA" in k else None,
            "hbm_bytes_per_launch": out.get("traffic", {}).get("hbm_bytes_per_launch"),
            "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (summed over SIMDs / XCDs), SQ_INSTS_*; FETCH_SIZE / "
                      "WRITE_SIZE; separate passes, tag " + tag}
    out["mfma"] = mfma
    json.dump(mfma, open(os.path.join(dst, "mfma_latest.json"), "w"), indent=1)
b = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(b) and os.path.getsize(b):
    out["bench_line_under_profiler"] = json.loads(open(b).read())
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out.get("traffic", {}), indent=1))
print(json.dumps(out.get("valu", {}), indent=1))
print("kernels:", list(summary))
