#!/usr/bin/env python3
"""Turn the gpurun_out/<tag>.* files of tools/profile_round.sh into the summaries kept under profiles/.

  python tools/summarize_profile.py r02_v6

profiles/<tag>_bench.json, _bench_ssl.json          the bench lines
profiles/<tag>_bench_kernel_stats.csv               rocprofv3 --stats of `bench.py --steps 10 --warmup 3`
profiles/<tag>_net_forward_B<boards>_kernel_stats.csv  rocprofv3 --stats of tools/bench_net.py <boards>
profiles/<tag>_conv_pmc_<counter>.csv               per-dispatch counters of the dominant kernel (trimmed columns)
profiles/conv_traffic.json                          hbm_bytes_per_launch etc. (read by bench.py for roofline.traffic)
"""
import csv, glob, json, os, shutil, statistics, sys

tag = sys.argv[1]
boards = int(sys.argv[2]) if len(sys.argv) > 2 else 24576        # boards per forward of the PMC passes (tools/profile_round.sh)
G = "gpurun_out"
P = "profiles"


def one(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    return f[0] if f else None


def copy(src_pat, dst):
    f = one(src_pat)
    if f:
        shutil.copy(f, os.path.join(P, dst))
    return f


for name in ("bench", "bench_ssl", "bench_sims1600", "bench_streams2"):
    f = os.path.join(G, f"{tag}.{name}.json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, f"{tag}_{name}.json"))
copy(f"{G}/{tag}.stats/**/*_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv")
copy(f"{G}/{tag}.netstats/**/*_kernel_stats.csv", f"{tag}_net_forward_B{boards}_kernel_stats.csv")

per_counter = {}
for d in sorted(glob.glob(f"{G}/{tag}.pmc_*")):
    if not os.path.isdir(d):
        continue
    cname = d.split(".pmc_")[1]
    f = one(f"{d}/**/*_counter_collection.csv")
    if not f:
        continue
    rows = list(csv.DictReader(open(f)))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count",
            "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    keep = [k for k in keep if rows and k in rows[0]]
    with open(os.path.join(P, f"{tag}_conv_pmc_{cname}.csv"), "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=keep)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in keep})
    for r in rows:
        per_counter.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))

if "FETCH_SIZE" in per_counter and "WRITE_SIZE" in per_counter:
    out = {"kernel": f"conv_zs_kernel<*> (3x3 320->320 implicit GEMM, zero padding skipped), B = {boards} boards per launch",
           "boards_per_launch": boards,
           "collected": f"tools/profile_round.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) "
                        f"--kernel-include-regex conv_zs_kernel -- python3 tools/bench_net.py {boards}; medians per kernel variant, "
                        "launch-weighted mean",
           "correction": "gfx950: FETCH_SIZE counts 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): fetch bytes = "
                         "2 x FETCH_SIZE KB; WRITE_SIZE KB taken as is",
           "per_variant": {}}
    tot_b = tot_n = 0
    for k, fv in per_counter["FETCH_SIZE"].items():
        wv = per_counter["WRITE_SIZE"].get(k, [0.0])
        fmb = 2 * statistics.median(fv) * 1024 / 1e6
        wmb = statistics.median(wv) * 1024 / 1e6
        out["per_variant"][k] = {"fetch_MB_corrected": round(fmb, 1), "write_MB": round(wmb, 1), "launches": len(fv)}
        tot_b += (fmb + wmb) * 1e6 * len(fv)
        tot_n += len(fv)
    out["hbm_bytes_per_launch"] = int(tot_b / max(tot_n, 1))
    out["algorithmic_bytes_per_launch"] = int(boards * 64 * 320 * 2 * 2 + 1843200)
    out["algorithmic_note"] = (f"plain/conv1: input {boards}*64*320*2 B + output the same + 1.84 MB weights; conv2 with tail: "
                               "+ x read + second output (twice the activation bytes)")
    sq = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
              "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_LDS"):
        if c in per_counter:
            sq[c] = {k: statistics.median(v) for k, v in per_counter[c].items()}
    if sq:
        out["sq_counters_median_per_launch"] = sq
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
            out["mfma_busy_over_gui_active"] = {
                k: round(sq["SQ_VALU_MFMA_BUSY_CYCLES"][k] / 1024.0 / (sq["GRBM_GUI_ACTIVE"][k] / 8.0), 3)
                for k in sq["SQ_VALU_MFMA_BUSY_CYCLES"] if k in sq["GRBM_GUI_ACTIVE"]}
            out["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs / GRBM_GUI_ACTIVE summed over 8 XCDs: share of the "
                                     "kernel's cycles in which a SIMD's matrix pipe is busy")
    old = os.path.join(P, "conv_traffic.json")
    if os.path.exists(old):
        prev = json.load(open(old))
        hist = prev.pop("history", {})
        if "r01" not in hist:
            hist["r01"] = prev
        out["history"] = hist
    json.dump(out, open(old, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "per_variant")}, indent=1))
    if "mfma_busy_over_gui_active" in out:
        print(out["mfma_busy_over_gui_active"])
