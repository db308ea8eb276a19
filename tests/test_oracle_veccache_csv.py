"""Pins the oracle's vector-granular cache (class cache2 of the reference, as dlrmtest2 drives it)
against the reference's own run artefacts dlrm_libtorch/build/LRU_RMC{1,2}_C2_B40_D200.csv: for every
batch and every table, the pages read by the lookups and the pages written by the reclaim must be
reproduced exactly.  Fixture: tests/golden/reference_lru_vec_trace.json (made by
tests/golden/extract_reference_csv.py)."""
import json
import os

import numpy as np
import pytest

from tests.refdata import reference_batches, reference_dataset

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_vec_trace.json")))


def run_trace(make_cache, cfg):
    """The train loop of src/dlrmtest2.cpp:542-603 reduced to its cache traffic: per batch the lookups
    of every table (emb::emb, :217-232), reclaim_emb_cache (:462-481), then save_emb for every table
    (update_parameter, :401-418).  make_cache(T, rows, D, cache_size) returns an object with
    get / reclaim / save / counters."""
    T, D, L = cfg["tables"], cfg["dim"], cfg["lookups"]
    B, N, c, rows = cfg["batch"], cfg["data_size"], cfg["cache_mult"], cfg["rows_per_table"]
    _, ids, _ = reference_dataset(T, L, B, N, rows)
    vc = make_cache(T, rows, D, c * L * B)               # cache_size = c * lookup_num * batch_size (:509)
    trace = []
    for epoch in range(cfg["epochs"]):
        for s, bs in reference_batches(N, B):
            req = [ids[s:s + bs, t * L:(t + 1) * L].reshape(-1) for t in range(T)]   # unsorted (:63-67, :128)
            for t in range(T):
                vc.get(t, req[t])
            reads = vc.counters("read", reset=True)
            vc.reclaim()
            writes = vc.counters("write", reset=True)
            for t in range(T):
                vc.save(t, req[t])
            trace.append({"read": reads.tolist(), "write": writes.tolist()})
    return trace


@pytest.mark.parametrize("name", sorted(GOLD))
def test_veccache_matches_reference_csv(oracle, name):
    cfg = GOLD[name]
    got = run_trace(lambda T, rows, D, cs: oracle.VectorCache(T, rows, D, cs), cfg)
    for b, (g, w) in enumerate(zip(got, cfg["batches"])):
        assert g == w, "batch %d" % b
    assert len(got) == len(cfg["batches"])


def test_veccache_fifo_and_repacking(oracle):
    vc = oracle.VectorCache(1, 4096, 32, cache_size=300)          # vpo = 32, ws_opt = 8: a buffer holds 256 vectors
    assert vc.vpo == 32
    first = np.arange(0, 4096, 8)                                  # 512 vectors, 4 per home page, 128 pages
    assert vc.get(0, first).tolist() == list(range(128))           # one page read per home page
    assert vc.counters("miss").tolist() == [512] and vc.counters("hit").tolist() == [0]
    assert vc.get(0, [0, 1, 8]).size == 0                          # vector 1 came along with its home page
    assert vc.fifo_order(0)[:3].tolist() == [0, 8, 16] and vc.list_len(0) == 513   # 1 is new to the list, 0 and 8 counted a hit
    res, lst, cnt, loc = vc.export(0)
    assert cnt[0] == 1 and cnt[8] == 1 and cnt[1] == 0 and lst[1] == 1 and res[2] == 1 and lst[2] == 0
    assert vc.reclaim() == 213                                     # 513 - 300 calls, oldest first
    assert vc.pending(0) == 213 and vc.counters("write").tolist() == [0]
    res, lst, cnt, loc = vc.export(0)
    assert res[0] == 0 and lst[0] == 0 and res[2] == 0            # 0 left; its unlisted neighbours were dropped
    assert res[1] == 1 and lst[1] == 1                             # listed neighbours stay
    vc.get(0, np.arange(1, 4096, 64))                              # 64 vectors: 63 new list entries (1 is listed)
    assert vc.reclaim() == 63
    assert vc.pending(0) == 20 and vc.counters("write").tolist() == [8] and vc.written_pages(0) == 8
    assert vc.page_count == -8
    res, lst, cnt, loc = vc.export(0)
    assert loc[0] == 128 and loc[8 * 31] == 128 and loc[8 * 36] == 129     # evicted together -> one new page
    # reading vector 0 again fetches the new page and brings the 31 others it was evicted with
    assert vc.get(0, [0]).tolist() == [128]
    res, lst, cnt, loc = vc.export(0)
    assert res[8] == 1 and lst[8] == 0 and res[8 * 31] == 1 and res[8 * 36] == 0
