"""Summarise a scripts/prof_bench.sh output directory: per-kernel avg duration (kernel trace)
and FETCH_SIZE / WRITE_SIZE per launch (PMC passes).  FETCH_SIZE is doubled for the gfx950
under-count of wide coalesced reads (MI355X_MICROARCH.md, HBM section); both raw and corrected
values are printed.  Counter unit: KiB (rocprofv3 derived metric: bytes/1024)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]   # [--merge ROWSxGPUS [--reset]] see the end of the file


def find(sub, pat):
    return sorted(glob.glob(os.path.join(d, sub, "**", pat), recursive=True))


out = {}
for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, d))
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")
        if "golap" in name or "nccl" in name.lower() or "rccl" in name.lower():
            print(f"  {name[:70]:70s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} avg_ns={row.get('AverageNs')} pct={row.get('Percentage')}")
            out.setdefault(name, {})["avg_ns"] = float(row.get("AverageNs") or 0)
            out[name]["calls"] = int(row.get("Calls") or 0)
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in find(sub, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != ctr:
                continue
            k = row.get("Kernel_Name", "")
            acc[k] += float(row.get("Counter_Value") or 0)
            cnt[k] += 1
    print(f"== {ctr} per launch (KiB raw)")
    for k in sorted(acc, key=lambda x: -acc[x]):
        if "golap" not in k:
            continue
        per = acc[k] / max(1, cnt[k])
        print(f"  {k[:70]:70s} launches={cnt[k]} raw_KiB={per:.0f} bytes={per*1024:.3e}")
        out.setdefault(k, {})[ctr + "_KiB_per_launch"] = per
# SQ pass: per-kernel sums over its launches; ratios are what matter (quad-cycle units, MI355X_MICROARCH.md counter table)
sq = defaultdict(lambda: defaultdict(float))
for f in find("pmc_sq", "*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "golap" in k:
            sq[k][row.get("Counter_Name")] += float(row.get("Counter_Value") or 0)
if sq:
    print("== SQ summary (share of wave cycles: parked at s_waitcnt/barrier | issue-stalled | issuing; LDS conflict share of LDS-active cycles)")
    for k in sorted(sq, key=lambda x: -sq[x].get("SQ_WAVE_CYCLES", 0)):
        c = sq[k]
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
        if c.get("SQ_WAVE_CYCLES", 0) < 1e6:
            continue
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0
        print(f"  {k[:70]:70s} wait_any={c.get('SQ_WAIT_ANY', 0) / wc:.3f} wait_inst={c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} "
              f"active_inst={c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} lds_conflict={c.get('SQ_LDS_BANK_CONFLICT', 0) / lds:.3f} "
              f"lds_active_per_busy={c.get('SQ_LDS_IDX_ACTIVE', 0) / (c.get('SQ_BUSY_CYCLES', 0) or 1.0):.3f}")
        out.setdefault(k, {})["sq"] = dict(c)
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)

# HBM traffic per launch for bench.py's roofline.traffic: FETCH_SIZE x 2 (gfx950 under-count of wide coalesced
# reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes.  Keyed by the library's profile names.
NAMES = [("k_part1_narrow<12, 12288", "radix_scatter_p1_wide"), ("k_part1_narrow<", "radix_scatter_p1"), ("k_part2_ring<", "radix_scatter_p2"),
         ("k_join_lds16<0", "join_build_probe_lds"), ("k_join_lds<0, true", "join_build_probe_lds"),
         ("k_scatter_big<1", "radix_scatter_p1_wide"), ("k_part2_wide<", "radix_scatter_p2_wide"), ("k_scatter_wc<2, true", "radix_scatter_p2_wide"), ("k_join_lds<0, false", "join_build_probe_lds_wide"),
         ("k_hist<1", "radix_hist_p1"),
         ("k_group_lds<", "hash_aggregate_lds"), ("k_agg_scatter_rsv<", "agg_partition_scatter"), ("k_agg_scatter_wc<", "agg_partition_scatter"), ("k_agg_scatter_big<", "agg_partition_scatter"),
         ("k_agg_scatter<", "agg_partition_scatter"),
         ("k_agg_hist<", "agg_partition_hist"), ("k_agg_bucket<", "hash_aggregate_lds_buckets"), ("k_agg_bucket(", "hash_aggregate_lds_buckets"),
         ("k_filter_mask", "filter_mask"), ("k_filter_count", "filter_count"), ("k_filter_write", "filter_compact"),
         ("k_reduce_partials", "filter_sum"), ("k_reduce_partials", "aggregate_ungrouped")]   # (one kernel serves both profile names)
traffic = {}
for k, v in out.items():
    for pat, prof in NAMES:   # first pattern that matches names the kernel; of several kernels under one name (a form that did the work and
                              # its stand-by twin, which returned at once) the one that moved the most bytes stands for the name
        if pat in k:
            if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
                f, w = v["FETCH_SIZE_KiB_per_launch"] * 1024, v["WRITE_SIZE_KiB_per_launch"] * 1024
                if prof not in traffic or 2 * f + w > traffic[prof]["hbm_bytes_per_launch"]:
                    traffic[prof] = {"kernel": k.split("(")[0], "fetch_size_raw_bytes": f, "read_bytes": 2 * f, "write_bytes": w,
                                     "hbm_bytes_per_launch": 2 * f + w, "rocprof_avg_ns": v.get("avg_ns")}
            if prof not in ("filter_sum",):
                break
json.dump(traffic, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print("== HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE)")
for k, v in traffic.items():
    print(f"  {k:28s} {v['hbm_bytes_per_launch']/1e9:8.2f} GB  (read {v['read_bytes']/1e9:.2f} + write {v['write_bytes']/1e9:.2f})  avg {v['rocprof_avg_ns']}")

# merge into profiles/pmc_traffic.json: usage  prof_summarize.py DIR --merge ROWSxGPUS [--reset]
# (bench.py only reports `traffic` while kernel_source_sha matches the sources it runs on; --reset drops entries of older sources)
if "--merge" in sys.argv:
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    cs = os.path.join(root, "gpu_olap_engine_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(cs, f), "rb").read())
    sha = h.hexdigest()[:16]
    size = sys.argv[sys.argv.index("--merge") + 1]
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        tj = {}
    if tj.get("kernel_source_sha") != sha or "--reset" in sys.argv:
        tj = {}
    tj["_comment"] = ("HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 per MI355X_MICROARCH.md HBM section, + WRITE_SIZE), one MI355X; "
                      "key = <profile kernel name>@<rows>x<n_gpus>; valid only for the kernel sources whose hash is kernel_source_sha (bench.py checks).")
    tj["kernel_source_sha"] = sha
    for k, v in traffic.items():
        if v["hbm_bytes_per_launch"] < 1e6:      # stand-by launches that returned at once
            continue
        v = dict(v)
        v["source"] = os.path.basename(os.path.normpath(d))
        tj[f"{k}@{size}"] = v
    json.dump(tj, open(path, "w"), indent=1)
    print("merged", len(traffic), "entries into", path, "sha", sha)
