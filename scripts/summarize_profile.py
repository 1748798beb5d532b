#!/usr/bin/env python3
"""Turn a rocprofv3 output tree (gpurun_out/prof_rNN, written by scripts/profile_*.sh) into the
small tracked files under profiles/: the kernel-trace stats table and one JSON with the PMC
counters of the traversal kernel, per launch.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes
of 16-B-per-lane reads, so  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024  bytes per launch.

    python3 scripts/summarize_profile.py gpurun_out/prof_r1 profiles/r01_coherent
"""
import collections
import csv
import glob
import json
import os
import sys

KERNEL = os.environ.get("RTK_PROFILE_KERNEL", "rtk_trace_kernel<0, false>")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    out = {"kernel": KERNEL, "source": src}
    # identity of the traversal kernels this was measured on: a hash over their machine code in the built library
    # (rtk_amd/kernel_id.py; bench.py drops the traffic figure when it differs). RTK_PROFILE_WORKLOAD names the workload.
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from rtk_amd import kernel_id
    wl = os.environ.get("RTK_PROFILE_WORKLOAD")
    if wl:
        out["workload"] = wl
        out["kernel_code_sha16"] = kernel_id.workload_kernel_sha16(os.path.join(root, "rtk_amd", "librtk_amd.so"), wl)
    out["rays_per_launch"] = int(os.environ.get("RTK_PROFILE_RAYS", str(1 << 24)))
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(dst + "_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
                if KERNEL in r["Name"]:
                    out["kernel_trace"] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                           "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    counters = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["dispatch"] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                                   "lds_block_size": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
        for k, v in agg.items():
            counters[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    out["pmc_per_launch"] = counters
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        out["hbm_traffic_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"]["mean"] + counters["WRITE_SIZE"]["mean"]) * 1024.0
        out["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024: KiB units, gfx950 FETCH_SIZE half-count correction (MI355X_MICROARCH.md, HBM)"
    if "TCC_HIT_sum" in counters and "TCC_MISS_sum" in counters:
        h, m = counters["TCC_HIT_sum"]["mean"], counters["TCC_MISS_sum"]["mean"]
        out["l2_hit_rate"] = h / (h + m)
    for name in ("bench_trace.json", "bench_fetch.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            try:
                out["bench_line_under_profiler"] = json.loads(open(p).read().strip().splitlines()[-1])
                break
            except Exception:
                pass
    with open(dst + "_pmc.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in out if k != "bench_line_under_profiler"}, indent=1))


if __name__ == "__main__":
    main()
