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

d = sys.argv[1]


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
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)

# HBM traffic per launch for bench.py's roofline.traffic: FETCH_SIZE x 2 (gfx950 under-count of wide coalesced
# reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes.  Keyed by the library's profile names.
NAMES = [("k_scatter_big<1", "radix_scatter_p1"), ("k_scatter<1", "radix_scatter_p1"), ("k_scatter_wc<2, true", "radix_scatter_p2"),
         ("k_scatter_wc<2>", "radix_scatter_p2"), ("k_scatter<2", "radix_scatter_p2"), ("k_hist<1", "radix_hist_p1"), ("k_join_lds<0", "join_build_probe_lds"),
         ("k_join_lds(", "join_build_probe_lds")]
traffic = {}
for k, v in out.items():
    for pat, prof in NAMES:
        if pat in k and prof not in traffic and "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            f, w = v["FETCH_SIZE_KiB_per_launch"] * 1024, v["WRITE_SIZE_KiB_per_launch"] * 1024
            traffic[prof] = {"kernel": k.split("(")[0], "fetch_size_raw_bytes": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w,
                             "hbm_bytes_per_launch": 2 * f + w, "avg_ns": v.get("avg_ns")}
json.dump(traffic, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print("== HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE)")
for k, v in traffic.items():
    print(f"  {k:24s} {v['hbm_bytes_per_launch']/1e9:8.2f} GB  (read {v['fetch_bytes_corrected_x2']/1e9:.2f} + write {v['write_bytes']/1e9:.2f})  avg {v['avg_ns']}")
