#!/usr/bin/python3
"""Summarise rocprofv3 CSV output (kernel-trace --stats and --pmc passes) into small text/JSON
files that are committed under profiles/.

    python tools/pmc_summary.py <rocprof_out_dir> [...more dirs] --out profiles/rNN_name

Writes <out>_kernel_stats.csv (copy of the --stats table, trimmed) and <out>_pmc.json with, per
kernel, the mean of every collected counter per dispatch, restricted to the largest dispatches of
each kernel (the finest level), plus HBM traffic per launch derived as the guide prescribes:
    read  bytes = 2 x FETCH_SIZE x 1024   (gfx950 reports half the bytes of coalesced streaming reads)
    write bytes =     WRITE_SIZE x 1024
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    stats_rows = []
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
            stats_rows = list(csv.reader(open(path)))
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                key = (r["Kernel_Name"], int(r["Grid_Size"]))
                counters[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                key = (r["Kernel_Name"], grid)
                durations[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if stats_rows:
        with open(args.out + "_kernel_stats.csv", "w") as fh:
            w = csv.writer(fh)
            for row in stats_rows:
                w.writerow([c if len(c) < 160 else c[:157] + "..." for c in row])
    # largest grid per kernel = finest level
    best = {}
    for (name, grid) in set(list(counters.keys()) + list(durations.keys())):
        if name.startswith("__amd_rocclr"):
            continue
        if name not in best or grid > best[name]:
            best[name] = grid
    def upper(vals):
        """levels that are launched with the same grid (one workgroup per CU) share a key: keep the finest level's
        dispatches = the upper cluster when the values are clearly bimodal"""
        lo, hi = min(vals), max(vals)
        if lo <= 0 or hi <= 3 * lo:
            return vals
        cut = (lo * hi) ** 0.5
        return [v for v in vals if v >= cut]

    out = {}
    for name, grid in sorted(best.items()):
        key = (name, grid)
        e = {"grid_size": grid}
        if durations.get(key):
            ds = upper(durations[key])
            e["dispatches"] = len(ds)
            e["avg_duration_us"] = round(sum(ds) / len(ds) / 1e3, 2)
        for cname, vals in counters.get(key, {}).items():
            vals = upper(vals)
            e[cname] = round(sum(vals) / len(vals), 2)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_per_launch"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
        out[name[:120]] = e
    with open(args.out + "_pmc.json", "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
