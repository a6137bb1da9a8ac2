"""Condenses rocprofv3 output under gpurun_out/prof_* into small tracked files under profiles/.

    python scripts/summarize_profiles.py r01

Writes profiles/<round>_scan_kernel_stats.csv (the --kernel-trace --stats summary, verbatim),
profiles/<round>_ingest_kernel_stats.csv, profiles/<round>_pmc.json (per-kernel counter averages)
and profiles/scan_filter_pmc.json (HBM traffic per launch of the dominant kernel, read by bench.py).
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(P, exist_ok=True)


def copy(src, dst):
    if os.path.exists(src):
        shutil.copy(src, dst)
        print("copied", dst)


copy(os.path.join(G, "prof_trace", "scan_kernel_stats.csv"), os.path.join(P, f"{tag}_scan_kernel_stats.csv"))
copy(os.path.join(G, "prof_ingest", "ingest_kernel_stats.csv"), os.path.join(P, f"{tag}_ingest_kernel_stats.csv"))

copy(os.path.join(G, "prof_small", "small_kernel_stats.csv"), os.path.join(P, f"{tag}_small_batch_kernel_stats.csv"))
f = os.path.join(G, "prof_small_pmc", "small_counter_collection.csv")
if os.path.exists(f):
    rows_ = list(csv.DictReader(open(f)))
    v = [float(r["Counter_Value"]) for r in rows_ if ("scan_tiles_i8r_kernel<0>" in r["Kernel_Name"] or "scan_tiles_i8r_kernel<0, false" in r["Kernel_Name"] or "scan_tiles_i8d_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE"]
    small_i8 = bool(v)
    if not v:
        v = [float(r["Counter_Value"]) for r in rows_ if "bf16n_kernel<1," in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    if v:
        fs = sum(v) / len(v)
        out = {"kernel": "scan_tiles_i8d_kernel / i8r (int8 shadow, one resident 128-query tile per CU, Q <= 128)" if small_i8 else
                         "scan_tiles_bf16n_kernel<COSINE, 2> (narrow filter, Q <= 64)", "rows_per_gpu": 12_500_000,
               "dim": 768, "queries": 64, "fetch_size_kib": fs, "hbm_bytes_per_launch": 2.0 * fs * 1024.0,
               "algorithmic_bytes_per_launch": 12_500_000 * 768 * (1 if small_i8 else 2) * 63 / 64,
               "correction": "2x (gfx950 FETCH_SIZE halves wide coalesced reads)", "launches": len(v), "round": tag}
        json.dump(out, open(os.path.join(P, f"{tag}_small_batch_pmc.json"), "w"), indent=1)
        print(out)

ipmc = {}
for d in ("prof_ingest_pmc1", "prof_ingest_pmc2"):
    f = os.path.join(G, d, "ingest_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        ipmc.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if ipmc:
    isum = {}
    for k_, cs in ipmc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if "SQ_INSTS_VALU" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
            # a wave64 VALU instruction holds its SIMD's 16-lane ALU for 4 cycles; 1024 SIMDs; GRBM summed over 8 XCDs
            e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * 4.0 / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes"] = 2.0 * e["FETCH_SIZE"] * 1024.0
        isum[k_] = e
    json.dump({"meta": {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python scripts/ingest_bench.py --gib 100 --reps 2",
                        "notes": ["valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (kernel cycles x 1024 SIMDs): the share of the "
                                  "chip's integer VALU issue slots the kernel used while it ran (concurrent kernels share them)",
                                  "FETCH_SIZE in KiB; hbm_read_bytes applies the x2 gfx950 correction, which is calibrated for wide "
                                  "coalesced streams (cdc_candidates_w48: 108 GB for 107 GB of blobs) and over-counts the "
                                  "lane-per-message 16-byte loads of sha256_batch_kernel (198 GB reported for 107 GB read)",
                                  "kernels are serialised under counter collection; in the product the blob-digest kernel "
                                  "runs beside CDC + chunk hashing on a second stream"]},
               "kernels": isum}, open(os.path.join(P, f"{tag}_ingest_pmc.json"), "w"), indent=1)
    print("wrote", os.path.join(P, f"{tag}_ingest_pmc.json"))

pmc = {}
for d in ("prof_pmc1", "prof_pmc2", "prof_pmc3"):
    f = os.path.join(G, d, "scan_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        e = pmc.setdefault(name, {}).setdefault(r["Counter_Name"], [])
        e.append(float(r["Counter_Value"]))
summary = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in pmc.items()}
meta = {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline --no-ingest (separate passes: FETCH_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                   "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE)",
        "notes": ["FETCH_SIZE is in KiB; on gfx950 it under-reports wide coalesced reads by 2x "
                  "(MI355X_MICROARCH.md, HBM section): hbm_bytes = 2 * FETCH_SIZE * 1024",
                  "GRBM_GUI_ACTIVE is summed over the 8 XCDs",
                  "SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16 (64 per "
                  "v_mfma_f32_32x32x2_f32, 16 per v_mfma_i32_16x16x64_i8), summed over 1024 SIMDs"]}
json.dump({"meta": meta, "kernels": summary}, open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
print("wrote", os.path.join(P, f"{tag}_pmc.json"))

filt = ([k for k in summary if ("scan_tiles_i8d_kernel" in k or "scan_tiles_i8r_kernel<0, false, true, 70" in k)] or     # the shipped sweep (NOT the sample pass <..., 0, true>)
        [k for k in summary if "scan_tiles_i8r_kernel<0>" in k or ("scan_tiles_i8r_kernel<0, false" in k and not k.rstrip(">").endswith("true"))] or [k for k in summary if "scan_tiles_i8h_kernel<1" in k] or [k for k in summary if "scan_tiles_i8_kernel<1" in k] or [k for k in summary if "bf16p_kernel" in k] or [k for k in summary if "bf16s_kernel<1, 0" in k] or [k for k in summary if "bf16k32_kernel<1, 0" in k]
        or [k for k in summary if "bf16v2_kernel<1, 0" in k] or [k for k in summary if "scan_tiles_kernel<1, 0>" in k])
if filt and "FETCH_SIZE" in summary[filt[0]]:
    fs = summary[filt[0]]["FETCH_SIZE"]["mean"]
    is_i8 = "scan_tiles_i8" in filt[0]
    out = {"kernel": filt[0], "rows_per_gpu": 12_500_000, "dim": 768, "queries": 1024, "bf16": is_i8 or "bf16" in filt[0],
           "passes": 3 if "bf16v2_kernel<1, 0, 3" in filt[0] else (1 if (is_i8 or "bf16" in filt[0]) else 0),
           "shadow": is_i8 or "bf16s_kernel" in filt[0] or "bf16p_kernel" in filt[0], "i8": is_i8,
           "fetch_size_kib": fs, "hbm_bytes_per_launch": 2.0 * fs * 1024.0,
           "correction": "2x (gfx950 FETCH_SIZE halves wide coalesced reads)", "round": tag}
    k2 = summary[filt[0]]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k2 and "GRBM_GUI_ACTIVE" in k2:
        out["mfma_busy_frac"] = k2["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (k2["GRBM_GUI_ACTIVE"]["mean"] / 8.0 * 1024.0)
    json.dump(out, open(os.path.join(P, "scan_filter_pmc.json"), "w"), indent=1)
    print(out)


# BASELINE config 2: kernel stats of its step + the sweep's HBM bytes per launch
copy(os.path.join(G, "prof_c2", "c2_kernel_stats.csv"), os.path.join(P, f"{tag}_config2_kernel_stats.csv"))
f = os.path.join(G, "prof_c2_pmc", "c2_counter_collection.csv")
if os.path.exists(f):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "scan_tiles_i8d_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    if v:
        fs = sum(v) / len(v)
        out = {"kernel": "scan_tiles_i8d_kernel", "rows": 1_000_000, "dim": 384, "queries": 256, "fetch_size_kib": fs, "hbm_bytes_per_launch": 2.0 * fs * 1024.0,
               "algorithmic_bytes_per_launch": 378101888, "correction": "2x (gfx950 FETCH_SIZE halves wide coalesced reads)", "launches": len(v), "round": tag}
        json.dump(out, open(os.path.join(P, f"{tag}_config2_pmc.json"), "w"), indent=1)
        print(out)


# the product-quantised engine's leg and the non-uniform legs: kernel stats; LDS counters of the ADC scan
copy(os.path.join(G, "prof_pq", "pq_kernel_stats.csv"), os.path.join(P, f"{tag}_pq_kernel_stats.csv"))
copy(os.path.join(G, "prof_dist", "dist_kernel_stats.csv"), os.path.join(P, f"{tag}_non_uniform_kernel_stats.csv"))
f = os.path.join(G, "prof_pq_pmc", "pq_counter_collection.csv")
if os.path.exists(f):
    acc_ = {}
    for r in csv.DictReader(open(f)):
        if "pq_adc_filter_kernel" in r["Kernel_Name"] and ", 2>" in r["Kernel_Name"]:
            acc_.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if acc_:
        out = {c: sum(v) / len(v) for c, v in acc_.items()}
        out["launches"] = max(len(v) for v in acc_.values())
        if out.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict_frac"] = out.get("SQ_LDS_BANK_CONFLICT", 0.0) / out["SQ_LDS_IDX_ACTIVE"]
        json.dump({"kernel": "pq_adc_filter_kernel<.., 4, 2>", "round": tag, "counters": out}, open(os.path.join(P, f"{tag}_pq_pmc.json"), "w"), indent=1)
        print(out)
