#!/usr/bin/env python3
"""Turns the page-traffic logs of the reference-shaped driver into hit rates.

    python tools/cache_trace.py pages  LRU_BASE_*.csv [--tables T]     # the reference's schema
    python tools/cache_trace.py trace  run.trace.csv                   # adapter/bin/dlrmtest_amd --trace

`pages` reads 'Time';'Channel';'PU';'Read';'Write' files (src/dlrmtest.cpp:448, written by the reference and by
adapter/bin/dlrmtest_amd): a batch is one block of read rows (one per table, src/dlrmtest.cpp:162-167) followed
by the write rows of the reclaim after it (:404-411).  It prints pages read / written per batch.
`trace` reads the driver's extended log, which also has the lookups that hit and missed, and prints the read
hit rate (lookups served from the cache), the page hit rate (pages wanted that were resident) and the share of
evicted pages per batch and in total."""
import argparse
import csv
import json
import sys


def read_pages(path, tables=None):
    """tables: read rows per batch (one per table).  Without it a batch ends where write rows are
    followed by read rows, which merges batches that were not followed by a reclaim."""
    batches, cur, mode, n_read = [], None, None, 0
    with open(path) as f:
        next(f)
        for line in f:
            p = line.strip().split(";")
            if len(p) < 4:
                continue
            is_read = p[3] != ""
            if cur is None or (is_read and (mode == "w" or (tables and n_read == tables))):
                cur = {"time": float(p[0]), "read": 0, "write": 0}
                batches.append(cur)
                n_read = 0
            mode = "r" if is_read else "w"
            n_read += 1 if is_read else 0
            if is_read:
                cur["read"] += int(p[3])
            elif len(p) > 4 and p[4] != "":
                cur["write"] += int(p[4])
    return batches


def read_trace(path):
    rows = []
    with open(path) as f:
        r = csv.reader(f, delimiter=";")
        head = [h.strip("'") for h in next(r)]
        for rec in r:
            rows.append(dict(zip(head, rec)))
    out = {}
    for rec in rows:
        k = (int(rec["Epoch"]), int(rec["Batch"]))
        d = out.setdefault(k, {"epoch": k[0], "batch": k[1], "hit": 0, "miss": 0, "read": 0, "write": 0, "evicted": 0,
                               "resident": 0})
        for name in ("Hit", "Miss", "Read", "Write", "Evicted", "Resident"):
            d[name.lower()] += int(rec[name])
    res = []
    for k in sorted(out):
        d = out[k]
        n = d["hit"] + d["miss"]
        d["read_hit_rate"] = d["hit"] / n if n else None
        res.append(d)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["pages", "trace"])
    ap.add_argument("path")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--tables", type=int, default=None, help="pages: tables (read rows) per batch")
    a = ap.parse_args()
    if a.kind == "pages":
        b = read_pages(a.path, a.tables)
        if a.json:
            print(json.dumps(b))
            return
        print("batch  pages_read  pages_written")
        for i, x in enumerate(b):
            print("%5d  %10d  %13d" % (i, x["read"], x["write"]))
        print("total  %10d  %13d" % (sum(x["read"] for x in b), sum(x["write"] for x in b)))
    else:
        b = read_trace(a.path)
        if a.json:
            print(json.dumps(b))
            return
        print("epoch batch  lookups  read_hit_rate  pages_read  pages_written  evicted  resident")
        for x in b:
            n = x["hit"] + x["miss"]
            print("%5d %5d  %7d  %13.4f  %10d  %13d  %7d  %8d" % (x["epoch"], x["batch"], n, x["read_hit_rate"] or 0.0,
                                                                 x["read"], x["write"], x["evicted"], x["resident"]))
        hit, miss = sum(x["hit"] for x in b), sum(x["miss"] for x in b)
        print("total read hit rate %.4f over %d lookups; %d pages read, %d evicted" %
              (hit / max(hit + miss, 1), hit + miss, sum(x["read"] for x in b), sum(x["evicted"] for x in b)))


if __name__ == "__main__":
    main()
