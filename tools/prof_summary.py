#!/usr/bin/env python3
"""Boils one rocprofv3 output directory down to what is kept under profiles/.

    python tools/prof_summary.py <rocprof_dir> <out_prefix>

--kernel-trace --stats runs: copies the per-kernel summary (<out_prefix>_kernel_stats.csv: calls, total / average /
min / max duration per kernel).  --pmc runs: sums every counter per kernel over its dispatches
(<out_prefix>_pmc.json: kernel -> counter -> {dispatches, sum, per_dispatch}; FETCH_SIZE / WRITE_SIZE are in KiB
as rocprofv3 reports them).  The per-dispatch traces are not kept."""
import collections
import csv
import glob
import json
import shutil
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    for f in glob.glob(src + "/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, prefix + "_kernel_stats.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0]))
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"][:160]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if acc:
        out = {k: {c: {"dispatches": n, "sum": s, "per_dispatch": s / n, "avg_ns_under_pmc": ns / n}
                   for c, (n, s, ns) in v.items()} for k, v in acc.items()}
        json.dump(out, open(prefix + "_pmc.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
