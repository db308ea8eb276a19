#!/usr/bin/env python3
"""Cache oversubscription sweep: tables live in pinned host DRAM, an HBM page cache of a fraction of
their size sits in front, ids are Zipf(1.05).  Per batch: lookup (probe, fill misses over the host
link, translate ids to slot rows), EmbeddingBag forward and fused sparse SGD update on the slot pool,
reclaim down to the capacity.  Reports hit rate, samples/s and host->device GB/s per cache fraction
and page size (rows_per_page = 1: hot-row cache; 4096 B pages: the reference's granularity).

    python tools/cache_sweep.py [out.json] [--tables 8 --rows 4000000 --dim 128 --batch 65536 --steps 30]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import _capi, ops  # noqa: E402
from dlrm_amd.cache import EmbeddingCache  # noqa: E402
from dlrm_amd.data import zipf_ids  # noqa: E402


def run(T, rows, D, B, steps, warm, frac, rows_per_page, alpha, dist, wb="async"):
    total_pages = T * ((rows + rows_per_page - 1) // rows_per_page)
    cache_pages = int(total_pages * frac)
    headroom = min(total_pages, T * B)                       # a batch can bring in at most T*B new pages
    c = EmbeddingCache([rows] * T, D, cache_pages, cache_pages + headroom, rows_per_page=rows_per_page)
    c.write(seed=1)
    if wb == "async":       # victims are parked in HBM and written to the store beside the next lookup's misses
        c.set_async_writeback(headroom)
    gen = torch.Generator(device="cuda").manual_seed(5)
    batches = []

    def make_batch():
        if dist == "zipf":
            return torch.cat([zipf_ids(B, rows, alpha, gen, "cuda") for _ in range(T)]).to(torch.int32)
        return torch.randint(0, rows, (T * B,), generator=gen, device="cuda", dtype=torch.int32)

    toff = [t * B for t in range(T + 1)]
    off = torch.arange(T * B + 1, device="cuda", dtype=torch.int32)
    out = torch.empty(T * B, D, device="cuda")
    grad = torch.randn(T * B, D, device="cuda") * 1e-3
    ws = None

    def step(ids):
        nonlocal ws
        r = c.get_emb_vectors(ids, toff)
        ops.embbag_forward(c.pool, r, off, T * B, 1, None, _capi.POOL_SUM, True, out=out, out_ld=D)
        ws = ops.embbag_backward_update(c.pool, r, off, T * B, grad, 1, None, _capi.POOL_SUM, True, lr=0.01,
                                        workspace=ws)
        c.reclaim()

    # warm up with fresh ids until the cache has filled and reclaim is evicting (steady state)
    n_warm = 0
    while n_warm < warm or (c.state().page_count < cache_pages and n_warm < 600):
        step(make_batch())
        n_warm += 1
    batches = [make_batch() for _ in range(steps)]          # every timed step sees fresh ids
    torch.cuda.synchronize()
    c.counters(reset=True)
    c0 = c.counters()
    t0 = time.perf_counter()
    for i in range(steps):
        step(batches[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = c.counters()
    hit = int((c1["hit"] - c0["hit"]).sum())
    miss = int((c1["miss"] - c0["miss"]).sum())
    read_pages = int(c1["read"].sum())
    written = int(c1["write"].sum())
    st = c.state()
    page_bytes = rows_per_page * D * 4
    rec = {"cache_fraction": frac, "rows_per_page": rows_per_page, "writeback": wb, "ids": dist, "tables": T, "rows": rows, "dim": D,
           "batch": B, "table_GB": T * rows * D * 4 / 1e9, "cache_GB": cache_pages * page_bytes / 1e9,
           "hit_rate_lookups": hit / max(hit + miss, 1), "pages_read_per_batch": read_pages / steps,
           "pages_written_back_per_batch": written / steps, "samples_per_s": B * steps / dt,
           "ms_per_batch": dt * 1e3 / steps, "h2d_GBps": read_pages * page_bytes / dt / 1e9,
           "d2h_GBps": written * page_bytes / dt / 1e9, "warmup_batches": n_warm,
           "error_flags": int(st.error_flags)}
    c.close()
    return rec


def run_vec(T, rows, D, B, steps, frac, alpha, replay):
    """The vector-granular policy (the reference's cache2): cache_size rows per table are listed, evicted rows
    are repacked into new pages of the store.  replay: the timed batches repeat an earlier sequence (a second
    epoch over the same data: rows that were evicted together are asked for together again)."""
    from dlrm_amd.vcache import VectorCache
    vpo = 4096 // (D * 4)
    cap = max(int(rows * frac), 8 * vpo)
    slots = min(T * rows, T * (cap + 2 * B) * vpo)
    n_seq = steps
    fill = (cap + B - 1) // max(B // 2, 1) + 2                 # batches until every list is full (upper bound)
    log_pages = (2 * n_seq + fill + 8) * ((B + vpo - 1) // vpo) + 8
    c = VectorCache([rows] * T, D, cap, slots, log_pages)
    c.write(seed=1)
    gen = torch.Generator(device="cuda").manual_seed(5)

    def make_batch():
        return torch.cat([zipf_ids(B, rows, alpha, gen, "cuda") for _ in range(T)]).to(torch.int32)

    toff = [t * B for t in range(T + 1)]
    off = torch.arange(T * B + 1, device="cuda", dtype=torch.int32)
    out = torch.empty(T * B, D, device="cuda")
    grad = torch.randn(T * B, D, device="cuda") * 1e-3
    ws = None

    def step(ids):
        nonlocal ws
        r = c.get_emb_vectors(ids, toff)
        ops.embbag_forward(c.pool, r, off, T * B, 1, None, _capi.POOL_SUM, True, out=out, out_ld=D)
        ws = ops.embbag_backward_update(c.pool, r, off, T * B, grad, 1, None, _capi.POOL_SUM, True, lr=0.01,
                                        workspace=ws)
        c.reclaim()
        c.restore(ids, toff)

    n_warm = 0
    while int(c.counters()["listed"].min()) < cap and n_warm < fill:
        step(make_batch())
        n_warm += 1
    seq = [make_batch() for _ in range(n_seq)]
    if replay:                                                 # first epoch over the sequence, untimed
        for ids in seq:
            step(ids)
    torch.cuda.synchronize()
    c.counters(reset=True)
    c0 = c.counters()
    t0 = time.perf_counter()
    for ids in seq:
        step(ids)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = c.counters()
    hit = int((c1["hit"] - c0["hit"]).sum())
    miss = int((c1["miss"] - c0["miss"]).sum())
    st = c.state()
    rec = {"policy": "vector FIFO + repacked pages (cache2)", "replayed_sequence": bool(replay), "cache_fraction": frac,
           "rows_per_page": vpo, "ids": "zipf", "tables": T, "rows": rows, "dim": D, "batch": B,
           "table_GB": T * rows * D * 4 / 1e9, "listed_rows_per_table": cap, "slot_pool_GB": slots * D * 4 / 1e9,
           "resident_rows": int(c1["resident"].sum()), "hit_rate_lookups": hit / max(hit + miss, 1),
           "pages_read_per_batch": int(c1["read"].sum()) / steps, "pages_written_per_batch": int(c1["write"].sum()) / steps,
           "samples_per_s": B * steps / dt, "ms_per_batch": dt * 1e3 / steps,
           "h2d_GBps": int(c1["read"].sum()) * 4096 / dt / 1e9, "d2h_GBps": int(c1["write"].sum()) * 4096 / dt / 1e9,
           "warmup_batches": n_warm, "error_flags": int(st.error_flags)}
    c.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out", nargs="?")
    ap.add_argument("--tables", type=int, default=8)
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=1.05)
    ap.add_argument("--fracs", default="0.5,0.25,0.125,0.0625")
    ap.add_argument("--wb", default="async", choices=["async", "sync"],
                    help="page-LRU policy: victims written back beside the next lookup (dlrm_cache_set_async_writeback) or inside the reclaim")
    ap.add_argument("--rpp", default="", help="rows per page, comma separated (default: 1 and the 4096-byte page)")
    ap.add_argument("--policy", default="lru", choices=["lru", "vec"],
                    help="lru: the page/row LRU cache; vec: the vector-granular policy over a log-structured store")
    a = ap.parse_args()
    res = []
    fracs = [float(x) for x in a.fracs.split(",")]
    if a.policy == "vec":
        for replay in (False, True):
            for frac in fracs:
                rec = run_vec(a.tables, a.rows, a.dim, a.batch, a.steps, frac, a.alpha, replay)
                res.append(rec)
                print(json.dumps(rec), flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
        return
    rpps = [int(x) for x in a.rpp.split(",")] if a.rpp else [1, 4096 // (a.dim * 4)]
    for rpp in rpps:
        for frac in fracs:
            rec = run(a.tables, a.rows, a.dim, a.batch, a.steps, a.warmup, frac, rpp, a.alpha, "zipf", a.wb)
            res.append(rec)
            print(json.dumps(rec), flush=True)
    if not a.rpp:
        rec = run(a.tables, a.rows, a.dim, a.batch, a.steps, a.warmup, 0.25, 1, a.alpha, "uniform", a.wb)
        res.append(rec)
        print(json.dumps(rec), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
