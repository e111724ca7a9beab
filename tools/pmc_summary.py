#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs, --kernel-trace only) into per-kernel HBM
traffic per launch, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact.

    python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json
"""
import collections, csv, json, sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k, v in f.items():
        ww = w.get(k, [0.0])
        fa, wa = sum(v) / len(v), sum(ww) / len(ww)
        out[k] = {"launches": len(v), "FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa,
                  "traffic_bytes_per_launch": 2.0 * fa * 1024.0 + wa * 1024.0}
    json.dump({"note": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE under-count corrected)", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:80]:80s} n={v['launches']:4d} traffic/launch={v['traffic_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
