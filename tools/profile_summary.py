"""Turn rocprofv3 CSV output into the JSON summaries bench.py reads (profiles/r02_mfma_utilisation.json, profiles/r02_traffic.json).

usage: python tools/profile_summary.py mfma <counter_collection.csv> <out.json> "<command line that was profiled>"
       python tools/profile_summary.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>"
Kernels are grouped into the families bench.py's roofline uses (faoctasr_last_route)."""
import collections
import csv
import json
import re
import sys

FAMILY = [("igemm_wino_kernel", "winograd"), ("igemm_patch_kernel", "patch"), ("igemm_gather_kernel", "gather_flat"), ("igemm_nm_kernel", "narrow"), ("igemm_bf16x3_kernel", "bf16x3"),
          ("conv_m1_fwd", "m1_head"), ("igemm_wgrad_kernel", "wgrad_flat"), ("wgrad_patch_kernel", "wgrad_patch"), ("wgrad_s1_kernel", "wgrad_s1"),
          ("wgrad_x3_kernel", "wgrad_x3"), ("conv_m1_wgrad", "m1_wgrad"), ("stem_dgrad", "stem_dgrad"), ("stem_wgrad", "stem_wgrad"), ("norm_", "norm"), ("haar_", "haar"), ("ssim_", "ssim")]


def family(name):
    for pat, fam in FAMILY:
        if pat in name:
            return fam
    return None


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


def read(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return agg, {k: len(v) for k, v in disp.items()}, dur


def mfma(path, out, cmd):
    agg, n, dur = read(path)
    kernels, fams = {}, collections.defaultdict(lambda: collections.defaultdict(float))
    for k, v in agg.items():
        g = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                     # summed over the 8 XCDs
        if g <= 0:
            continue
        e = {"launches": n[k], "avg_us": round(dur[k] / n[k], 1), "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024), 3),
             "lds_active_frac": round(v.get("SQ_LDS_IDX_ACTIVE", 0.0) / (g * 256), 3),
             "lds_conflict_of_active": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3),
             "waves_per_simd": round(v.get("SQ_WAVE_CYCLES", 0.0) * 4 / (g * 1024), 2),
             "wait_any_of_wave": round(v.get("SQ_WAIT_ANY", 0.0) / max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0), 3)}
        if e["mfma_busy_frac"] > 0 or family(k):
            kernels[k] = e
        f = family(k)
        if f:
            for c, x in v.items():
                fams[f][c] += x
            fams[f]["_n"] += n[k]
    fam_out = {}
    for f, v in fams.items():
        g = v["GRBM_GUI_ACTIVE"] / 8.0
        fam_out[f] = {"launches": int(v["_n"]), "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024), 3),
                      "lds_conflict_of_active": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3)}
    json.dump({"source": cmd, "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the share of SIMD cycles in which the "
               "matrix pipe executes (GRBM_GUI_ACTIVE is summed over the 8 XCDs and reads high on dispatches shorter than ~0.3 ms, so short kernels are "
               "under-stated); lds_active_frac = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs)", "families": fam_out, "kernels": kernels},
              open(out, "w"), indent=1)


def traffic(fetch_path, write_path, out, cmd):
    fa, fn, _ = read(fetch_path)
    wa, wn, _ = read(write_path)
    kernels, fams = {}, collections.defaultdict(lambda: collections.defaultdict(float))
    for k in sorted(set(fa) | set(wa)):
        fkib = fa.get(k, {}).get("FETCH_SIZE", 0.0) / max(fn.get(k, 1), 1)
        wkib = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
        f = family(k)
        if f is None and fkib + wkib < 1024:
            continue
        kernels[k] = {"launches": fn.get(k, wn.get(k, 0)), "fetch_kib_per_launch": round(fkib, 1), "write_kib_per_launch": round(wkib, 1)}
        if f:
            fams[f]["fetch"] += fa.get(k, {}).get("FETCH_SIZE", 0.0)
            fams[f]["write"] += wa.get(k, {}).get("WRITE_SIZE", 0.0)
            fams[f]["n"] += fn.get(k, 0)
    fam_out = {}
    for f, v in fams.items():
        n = max(v["n"], 1)
        raw = (v["fetch"] + v["write"]) * 1024 / n
        x2 = (2 * v["fetch"] + v["write"]) * 1024 / n
        fam_out[f] = {"launches": int(v["n"]), "fetch_kib_per_launch": round(v["fetch"] / n, 1), "write_kib_per_launch": round(v["write"] / n, 1),
                      "hbm_bytes_per_launch_raw": int(raw), "hbm_bytes_per_launch": int(x2)}
    json.dump({"source": cmd, "note": "per-launch averages; FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE reads 1/2 of the bytes of "
               "wide (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM section): hbm_bytes_per_launch = 2 x FETCH + WRITE is the corrected figure "
               "(an upper bound for kernels that also issue 4-byte loads), hbm_bytes_per_launch_raw the uncorrected one", "families": fam_out,
               "kernels": kernels}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
