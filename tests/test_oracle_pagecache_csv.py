"""Pins the oracle's page-LRU cache state machine against the reference's own
run artefacts (dlrm_libtorch/build/LRU_BASE_RMC{1,2,3}_C2_B40_D200.csv):
per-batch totals of pages read and pages written must be reproduced exactly.
Fixture: tests/golden/reference_lru_base_totals.json (made by
tests/golden/extract_reference_csv.py)."""
import json
import os

import numpy as np
import pytest

from tests.refdata import reference_batches, reference_dataset

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_base_totals.json")))


def run_trace(orc, cfg, on_batch=None):
    """The reference's train loop reduced to its cache traffic
    (src/dlrmtest.cpp:468-545)."""
    T, D, L = cfg["tables"], cfg["dim"], cfg["lookups"]
    B, N, c, rows = cfg["batch"], cfg["data_size"], cfg["cache_mult"], cfg["rows_per_table"]
    _, ids, _ = reference_dataset(T, L, B, N, rows)
    pc = orc.PageCache(T, rows, D, cache_size=c * T * L * B)
    trace = []
    for epoch in range(1, cfg["epochs"] + 1):
        for b, (s, bs) in enumerate(reference_batches(N, B)):
            for t in range(T):
                pc.get(t, np.sort(ids[s:s + bs, t * L:(t + 1) * L].reshape(-1)))
            reads = pc.counters("read", reset=True)
            pc.reclaim()
            writes = pc.counters("write", reset=True)
            if b + 1 == c and epoch == 1:
                pc.cache_size = pc.page_count
            trace.append({"read": int(reads.sum()), "write": int(writes.sum())})
    return trace


@pytest.mark.parametrize("name", ["rmc1", "rmc2", "rmc3"])
def test_pagecache_matches_reference_csv(oracle, name):
    cfg = GOLD[name]
    assert run_trace(oracle, cfg) == cfg["batches"]


def test_pagecache_lru_order_and_hits(oracle):
    pc = oracle.PageCache(1, 1024, 32, cache_size=4)  # vpo = 32
    assert pc.vpo == 32
    miss = pc.get(0, [0, 1, 40, 70, 71, 400])
    assert miss.tolist() == [0, 1, 2, 12]
    assert pc.lru_order(0).tolist() == [12, 2, 1, 0]
    miss = pc.get(0, [33, 500])
    assert miss.tolist() == [15]
    assert pc.lru_order(0).tolist() == [15, 1, 12, 2, 0]
    assert pc.counters("hit").tolist() == [1] and pc.counters("miss").tolist() == [7]
    assert pc.page_count == 5
    assert pc.reclaim() == 1  # one pass of ws_opt=8 pages: drops everything, counters move by 8
    assert pc.page_count == -3 and pc.counters("write").tolist() == [8]
    assert pc.lru_order(0).tolist() == []


def test_pagecache_unsorted_ids_double_read(oracle):
    # literal restatement: only CONSECUTIVE misses on one page are merged
    # (src/emb_cache.cpp:426-438)
    pc = oracle.PageCache(1, 1024, 32, cache_size=100)
    assert pc.get(0, [0, 40, 1]).tolist() == [0, 1, 0]
    assert pc.page_count == 3


def test_pagecache_slots(oracle):
    pc = oracle.PageCache(2, 1024, 32, cache_size=3, n_slots=16)
    pc.get(0, [0, 32, 64])
    pc.get(1, [0, 96])
    assert pc.slot_map(0)[:3].tolist() == [0, 1, 2] and pc.slot_map(1)[[0, 3]].tolist() == [3, 4]
    pc.reclaim()  # 5 > 3: one pass, 8 per table: all gone; slots return in eviction order
    assert (pc.slot_map(0) == -1).all() and (pc.slot_map(1) == -1).all()
    pc.get(1, [5 * 32])
    # ring was 0..15, head=5; freed 0,1,2,3,4 were appended after 15
    assert pc.slot_of(1, 5) == 5
