#!/usr/bin/env python3
"""Summarise gpurun_out/profiles_<tag>/<workload>/ (tools/collect_profiles.sh) into the committed evidence, ONE record per kernel the
bench line quotes:
   profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of that bench.py run
   profiles/<tag>_<workload>_pmc_summary.json   per-kernel averages of every PMC counter collected (separate passes)
   profiles/kernels_latest.json                 kernel name -> {code_sha, tag, stats, valu, mfma, traffic}: what bench.py's roofline
                                                fields read (only when `code_sha` matches the kernel sources it runs)
   profiles/<tag>_valu_static_mix_<workload>.json  the static opcode mix of the kernel's code object (prices the `other` VALU class)
Usage: python tools/summarize_profiles.py TAG [MICRO_TAG]

valu: the launch's vector instructions by class (SQ_INSTS_VALU_*: add/mul/fma/trans f32, int32, int64, cvt; `other` = SQ_INSTS_VALU minus
those [minus SQ_INSTS_MFMA where collected], priced by its static opcode mix) x the issue cycles per instruction of the class measured on
this part at 4 waves per SIMD (profiles/<micro>_valu_issue.json, tools/micro/valu_issue.hip) = the SIMD-cycles the launch needs at ideal
issue rates: floor_us = that / (1024 SIMDs x 2.4 GHz). A kernel that runs ONE wave per SIMD (the 13-state lane-per-rollout kernels at
K = 65536) cannot reach the 4-wave rates — that is what its `frac` then says.
mfma: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = the fraction of the kernel's cycles the matrix pipe was busy.
traffic: HBM bytes per MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950's FETCH_SIZE tallies 128-B requests at 64 B (x2)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
micro = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
import mppi_tf_amd  # noqa: E402
code_sha = mppi_tf_amd.build.source_sha()

latest_path = os.path.join(dst, "kernels_latest.json")
try:
    latest = json.load(open(latest_path))
except Exception:
    latest = {}
latest.setdefault("kernels", {})
latest["about"] = "per kernel: rocprofv3 evidence behind bench.py's roofline fields (tools/collect_profiles.sh + tools/summarize_profiles.py)"

mfile = os.path.join(dst, micro + "_valu_issue.json")
cyc = {r["op"]: r["cyc_per_inst_per_simd"] for r in json.load(open(mfile)) if r["waves_per_simd_median"] == 4} if os.path.exists(mfile) else None
KIND_PRICE = None
if cyc:
    KIND_PRICE = {"other:mov": cyc["v_mov_b32"], "other:bitop3": cyc["v_bitop3_b32"], "other:dpp_f32": cyc["v_add_f32_dpp quad_perm"],
                  "other:dpp_mov": cyc["v_mov_b32_dpp row_mirror"], "other:permlane_swap": cyc["v_mov_b32_dpp row_mirror"],
                  "other:cndmask": cyc["v_cndmask_b32_e64 (mask in s[44:45])"], "other:lane": cyc["v_readlane_b32"],
                  "other:minmax": cyc["v_max_f32"], "other:cmp": cyc["v_xor_b32"], "other:bitfield3": cyc["v_bitop3_b32"],
                  "other:misc": cyc["v_bitop3_b32"]}


def stats_of(csv_path):
    """kernel name -> {calls, avg_us, min_us, total_pct} from a rocprofv3 kernel_stats.csv"""
    out = {}
    for row in csv.DictReader(open(csv_path)):
        name = row["Name"].split("(")[0].replace("void ", "")
        out[name] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3,
                     "percent": float(row["Percentage"])}
    return out


for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    if not os.path.isdir(wdir):
        continue
    wl = os.path.basename(wdir)
    stats = {}
    # a re-collection merges its files beside an earlier one's (one <pid>_ prefix per run): the newest of a directory is the one that counts
    sfiles = sorted(glob.glob(os.path.join(wdir, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if sfiles:
        shutil.copy(sfiles[-1], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
        stats = stats_of(sfiles[-1])
    summary = collections.defaultdict(dict)
    for pdir in sorted(glob.glob(os.path.join(wdir, "pmc_*"))):
        pfiles = sorted(glob.glob(os.path.join(pdir, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not pfiles:
            continue
        f = pfiles[-1]
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "mppi::" not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, cs in acc.items():
            for c, v in cs.items():
                summary[name][c] = sum(v) / len(v)
                summary[name]["launches_" + c] = len(v)
    out = {"tag": tag, "workload": wl, "kernels": summary, "kernel_stats": stats}
    b = os.path.join(wdir, "bench_under_profiler.json")
    wl_sha = code_sha
    if os.path.exists(b) and os.path.getsize(b):
        try:
            out["bench_line_under_profiler"] = json.loads(open(b).read())
            # the kernel sources the profiled run itself executed (bench.py's roofline.code_sha), not the ones in the tree now
            wl_sha = out["bench_line_under_profiler"].get("roofline", {}).get("code_sha", code_sha)
        except ValueError:
            pass
    out["code_sha"] = wl_sha
    json.dump(out, open(os.path.join(dst, "%s_%s_pmc_summary.json" % (tag, wl)), "w"), indent=1, sort_keys=True)

    for kname, k in summary.items():
        if "k_rollout" not in kname and "k_step" not in kname:
            continue
        rec = {"code_sha": wl_sha, "tag": "%s_%s" % (tag, wl), "stats": stats.get(kname)}
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            rec["traffic"] = {"FETCH_SIZE_KiB_raw": k["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": k["WRITE_SIZE"],
                              "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
                              "hbm_bytes_per_launch": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0}
        n_mfma = k.get("SQ_INSTS_MFMA", 0.0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
            rec["mfma"] = {"mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (k["GRBM_GUI_ACTIVE"] / 8.0),
                           "mfma_insts_per_launch": n_mfma,
                           "other_vector_insts_per_launch": (k["SQ_INSTS_VALU"] - n_mfma) if "SQ_INSTS_VALU" in k else None,
                           "other_vector_insts_per_mfma": ((k["SQ_INSTS_VALU"] - n_mfma) / n_mfma) if "SQ_INSTS_VALU" in k and n_mfma else None}
        if "SQ_INSTS_VALU_MUL_F32" in k and cyc:
            classes = {"add_f32": k["SQ_INSTS_VALU_ADD_F32"], "mul_f32": k["SQ_INSTS_VALU_MUL_F32"], "fma_f32": k["SQ_INSTS_VALU_FMA_F32"],
                       "trans_f32": k["SQ_INSTS_VALU_TRANS_F32"], "int32": k["SQ_INSTS_VALU_INT32"], "int64": k["SQ_INSTS_VALU_INT64"],
                       "cvt": k["SQ_INSTS_VALU_CVT"]}
            classes["other"] = k["SQ_INSTS_VALU"] - sum(classes.values()) - n_mfma  # (SQ_INSTS_VALU counts the MFMAs too)
            mixfile = os.path.join(dst, "%s_valu_static_mix_%s.json" % (tag, wl))
            mix, static_all = {}, {}
            try:
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_static_mix.py"), kname.replace("mppi::", ""), mixfile],
                                      stdout=subprocess.DEVNULL)
                static_all = json.load(open(mixfile))["classes"]
                mix = {c: n for c, n in static_all.items() if c.startswith("other:") and c != "other:dpp_f32"}  # (DPP adds are counted, and priced, in add_f32)
            except Exception as e:  # the static mix only refines the price of `other`
                print("static mix of %s unavailable: %s" % (kname, e))
            other_price = (sum(n * KIND_PRICE.get(c, cyc["v_bitop3_b32"]) for c, n in mix.items()) / max(1, sum(mix.values()))) if mix else cyc["v_bitop3_b32"]
            price = {"add_f32": cyc["v_add_f32"], "mul_f32": cyc["v_mul_f32"], "fma_f32": cyc["v_fma_f32"],
                     "trans_f32": (cyc["v_log_f32"] + cyc["v_sqrt_f32"] + cyc["v_sin_f32"]) / 3, "int32": cyc["v_xor_b32"],
                     "int64": cyc["v_mad_u64_u32 (+0)"], "cvt": cyc["v_cvt_f32_u32"], "other": other_price}
            # packed fp32 (r05): the add / mul / fma counters tally a v_pk_* like a plain instruction (profiles/r05_valu_counter_classes.txt),
            # so each of the three classes is priced at the static plain : packed share of the kernel's code object (DPP adds sit in add_f32 too)
            pk_share = {}
            for c_, pk_op in (("add_f32", "v_pk_add_f32"), ("mul_f32", "v_pk_mul_f32"), ("fma_f32", "v_pk_fma_f32")):
                n_pk = static_all.get("pk_" + c_, 0)
                n_dpp = static_all.get("other:dpp_f32", 0) if c_ == "add_f32" else 0
                n_plain = static_all.get(c_, 0)
                tot = n_pk + n_dpp + n_plain
                if tot and pk_op in cyc:
                    price[c_] = (n_plain * price[c_] + n_pk * cyc[pk_op] + n_dpp * cyc["v_add_f32_dpp quad_perm"]) / tot
                    pk_share[c_] = {"plain": n_plain, "packed": n_pk, "dpp": n_dpp}
            rec["valu_static_plain_packed"] = pk_share
            rec["valu"] = {"simds": 1024, "clock_mhz": 2400.0, "insts_per_launch": classes, "cycles_per_inst": price,
                           "other_static_mix": mix, "active_quad_cycles_per_launch": k.get("SQ_ACTIVE_INST_VALU"),
                           "waves_per_launch": k.get("SQ_WAVES"),
                           "source": "instructions: rocprofv3 --pmc SQ_INSTS_VALU* (own pass); cycles: profiles/%s_valu_issue.json at 4 waves per SIMD" % micro}
            total = sum(classes[c] * price[c] for c in classes)
            rec["valu"]["floor_us"] = total / (1024 * 2400.0)
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if c in k:
                rec.setdefault("sq", {})[c] = k[c]
        latest["kernels"][kname] = rec
        print("%-50s %-12s stats %s  floor_us %s  mfma_busy %s" % (kname, wl, (rec["stats"] or {}).get("avg_us"), (rec.get("valu") or {}).get("floor_us"),
                                                                  (rec.get("mfma") or {}).get("mfma_busy_frac")))
    # the pre-launched instances (k_step_pc<.., 4 | 5>, MPPI_TUNE_PRELAUNCH): they run in the --kernel-trace pass only — the counter passes serialise
    # dispatches, the two-stream pipeline cannot form there and bench.py leaves it out (--no-prelaunched) — so their record is the stats alone
    for kname, st in stats.items():
        if "k_step_pc" in kname and kname.rstrip(">").rsplit(", ", 1)[-1] in ("4", "5") and kname not in summary:
            latest["kernels"][kname] = {"code_sha": wl_sha, "tag": "%s_%s" % (tag, wl), "stats": st,
                                        "note": "the pre-launched step (MPPI_TUNE_PRELAUNCH, opt-in): the duration INCLUDES the wait for U' of the previous step - resident early is "
                                                "the point; no counters: rocprofv3 --pmc serialises dispatches and the pipeline cannot form there"}
            print("%-50s %-12s stats %s  (pre-launched: kernel-trace only)" % (kname, wl, st.get("avg_us")))
# one source hash per file: entries of earlier sources go (bench.py reads an entry only when its code_sha is the running sources' anyway)
for kname in [k_ for k_, v_ in latest["kernels"].items() if v_.get("code_sha") != code_sha]:
    print("dropped (older sources): %s" % kname)
    del latest["kernels"][kname]
json.dump(latest, open(latest_path, "w"), indent=1, sort_keys=True)
