#!/usr/bin/env python3
"""profiles/counters.json and profiles/hbm_traffic.json from the counter passes of tools/profile_r6.sh.

usage: tools/make_counters.py gpurun_out/<tag> [--merge]   (reads <tag>/pmc_<workload>/pmc_*/p_counter_collection.csv;
       --merge: keep the entries of the existing files (same library version) for workloads <tag> has no passes of)
Per workload the dominant kernel's mean counters per launch become:
  cycles            GRBM_GUI_ACTIVE / 8 XCDs (the launch's duration in shader clocks)
  valu_busy         SQ_INSTS_VALU x 2 / (1024 SIMDs x cycles): a wave64 VALU instruction issues over two cycles on gfx950's SIMD-32
  ta_busy, td_busy  TA_TA_BUSY_sum, TD_TD_BUSY_sum / (256 CUs x cycles); td_tc_stall likewise (the data-return unit waiting on the L2)
  lds_busy          SQ_LDS_IDX_ACTIVE / (256 x cycles), lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wave_time         shares of SQ_WAVE_CYCLES: active / issue-stall / parked on a counter
  l2_line_requests  TCP_TCC_READ_REQ_sum, l2_latency_cycles = TCP_TCC_READ_REQ_LATENCY_sum / requests
  hbm_bytes         FETCH_SIZE KiB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024
Both files are stamped with the library version: bench.py drops them when daco_version() differs."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# workload directory -> (kernel substring, key in counters.json / hbm_traffic.json)
WORKLOADS = {
    "headline": ("tsp_scan32_kernel", "tsp500_a512_b64_scan"),
    "scan_sparse": ("scan_sparse_kernel<2, false, 4, false>", "tsp500_a512_b64_scan_sparse"),
    "c5_sparse": ("scan_sparse_kernel<4, false, 8, false>", "tsp1000_a2048_b64_scan_sparse"),
    "b1_lds_heads": ("scan_sparse_kernel<2, false, 4, true>", "tsp500_a512_b1_scan_sparse"),
    "deposit_heads": ("deposit_rows_kernel<true, 1, true>", "tsp500_a512_b64_update_heads"),
    "race": ("tsp_sample_kernel", "tsp500_a512_b64_race"),
    "race_head": ("scan_sparse_kernel<2, true, 4, false>", "tsp500_a512_b64_race_head"),
    "c2": ("scan16_kernel", "tsp100_a512_b256_scan"),
    "c4": ("scan16_kernel", "cvrp100_a512_b256_scan"),
    "c5": ("tsp_scan32_kernel", "tsp1000_a2048_b64_scan"),
    "nls": ("nls_kernel", "nls500_a256_b64"),
    "gnn": ("gnn_fused2_layer_kernel<false, false>", "gnn_fused2_layer_tsp500_k50_b64"),
    "cvrp_ls": ("cvrp_ls_kernel", "cvrp_ls_100_a512_b16"),
    "hgs_ls": ("hgs_ls_kernel", "hgs_ls_100_a512_b64"),
}


def kernel_means(d, needle):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    out_dir = sys.argv[1]
    from deepaco_amd import _lib
    version = _lib.ABI_VERSION
    counters = {"daco_version": version,
                "source": "profiles/r06_pmc_*.txt (tools/profile_r6.sh + tools/make_counters.py: rocprofv3 --pmc, one pass per counter "
                          "group, mean per launch of the workload's dominant kernel, all of this library version)"}
    traffic = {"daco_version": version,
               "source": "profiles/r06_pmc_*.txt (tools/profile_r6.sh: FETCH_SIZE KiB x 1024 x 2 [gfx950 correction] + WRITE_SIZE KiB x "
                         "1024, mean per launch of the dominant kernel)"}
    if "--merge" in sys.argv[2:]:
        for name, cur in (("counters.json", counters), ("hbm_traffic.json", traffic)):
            old = json.load(open(os.path.join(ROOT, "profiles", name)))
            if old.get("daco_version") == version:
                cur.update({k: v for k, v in old.items() if k not in ("daco_version", "source")})
    for wl, (needle, key) in WORKLOADS.items():
        m = kernel_means(os.path.join(out_dir, "pmc_" + wl), needle)
        if "GRBM_GUI_ACTIVE" not in m:
            print("no counters for", wl)
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        c = {"kernel": needle, "cycles": cyc}
        if "SQ_INSTS_VALU" in m:
            c["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
            c["valu_busy"] = m["SQ_INSTS_VALU"] * 2 / (1024 * cyc)
            c["salu_insts_per_launch"] = m.get("SQ_INSTS_SALU")
        if "SQ_ACTIVE_INST_VALU" in m:
            # cycles the SIMDs spent executing VALU instructions (the counter ticks every four cycles, like SQ_WAVE_CYCLES): the
            # measure for float64-heavy kernels, whose instructions take four cycles and not the two assumed above
            c["valu_active_cycles"] = m["SQ_ACTIVE_INST_VALU"] * 4
            c["valu_active"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
        if "TA_TA_BUSY_sum" in m:
            c["ta_busy"] = m["TA_TA_BUSY_sum"] / (256 * cyc)
            c["td_busy"] = m["TD_TD_BUSY_sum"] / (256 * cyc)
            c["td_tc_stall"] = m["TD_TC_STALL_sum"] / (256 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in m:
            c["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
            c["lds_conflict"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m["SQ_LDS_IDX_ACTIVE"] else 0.0
        if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
            w = m["SQ_WAVE_CYCLES"]
            c["wave_time"] = {"active": m["SQ_ACTIVE_INST_ANY"] / w, "issue_stall": m["SQ_WAIT_INST_ANY"] / w, "parked": m["SQ_WAIT_ANY"] / w}
        if "TCP_TCC_READ_REQ_sum" in m:
            c["l2_line_requests"] = m["TCP_TCC_READ_REQ_sum"]
            if m["TCP_TCC_READ_REQ_sum"]:
                c["l2_latency_cycles"] = m.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / m["TCP_TCC_READ_REQ_sum"]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            c["hbm_bytes"] = m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024
            traffic[key] = c["hbm_bytes"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and m["TCP_TOTAL_CACHE_ACCESSES_sum"] and "TCP_TCC_READ_REQ_sum" in m:
            # the CU's vector L1: the share of its accesses that did not go on to the L2 as read requests
            c["l1_accesses"] = m["TCP_TOTAL_CACHE_ACCESSES_sum"]
            c["l1_hit_rate"] = max(0.0, 1.0 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"])
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            c["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        counters[key] = c
        print(key, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items() if k != "wave_time"})
    # the GNN forward as a whole: every gnn_* launch of a forward (tools/run_gnn_batch.py ... 3: three forwards per pass)
    tot = collections.defaultdict(float)
    for f in sorted(glob.glob(os.path.join(out_dir, "pmc_gnn", "pmc_*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "gnn_" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    if tot:
        traffic["gnn_tsp500_k50_b64"] = (tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024) / 3.0
    if "nls500_a256_b64" in counters and "l2_line_requests" in counters["nls500_a256_b64"]:
        traffic["nls500_a256_b64_l2_line_requests"] = counters["nls500_a256_b64"]["l2_line_requests"]
    json.dump(counters, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
