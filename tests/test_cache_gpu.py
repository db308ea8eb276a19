"""HBM page cache vs the CPU restatement of the reference's cache state
machine: page->slot maps, page_count, and the read / write / hit / miss
counters must be bit-exact after every call; evicted and flushed pages must
carry their updated rows back to the store.  Also replays the reference's RMC1
run and compares per-batch page traffic with the reference's own CSV."""
import json
import os

import numpy as np
import pytest
import torch

from tests.refdata import reference_batches, reference_dataset

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_base_totals.json")))


def sorted_batch(rng, rows_list, n_per_table):
    ids, offs = [], [0]
    for r, n in zip(rows_list, n_per_table):
        ids.append(np.sort(rng.integers(0, r, size=n)))
        offs.append(offs[-1] + n)
    return np.concatenate(ids).astype(np.int64), offs


def compare_state(cache, pc, T):
    st = cache.state()
    # flag 4 (a page of the current call was evicted) is a warning to the caller: it fires when
    # the capacity is smaller than one batch, which some of these configurations are on purpose
    assert st.error_flags & ~4 == 0
    assert st.page_count == pc.page_count
    assert st.ring_head == pc.L.orc_pagecache_ring_head(pc.h) and st.ring_tail == pc.L.orc_pagecache_ring_tail(pc.h)
    cnt = cache.counters()
    for name in ("read", "write", "hit", "miss"):
        assert np.array_equal(cnt[name], pc.counters(name)), name
    for t in range(T):
        got = cache.page_map(t).astype(np.int64)
        ref = pc.slot_map(t)                      # the oracle's tables are all padded to the largest one
        assert np.array_equal(got, ref[:got.size]) and (ref[got.size:] == -1).all(), "slot map of table %d" % t


# async_wb: victims parked in HBM and written to the store by the cache's own stream beside the next lookup
# (dlrm_cache_set_async_writeback); 40 pages: some reclaims fit the parking area, larger ones write back in the call
@pytest.mark.parametrize("async_wb", [0, 10 ** 4, 40], ids=["wb-in-reclaim", "wb-async", "wb-async-small-area"])
@pytest.mark.parametrize("dim,vpo,idx_t", [(32, None, torch.int64), (128, 1, torch.int32), (16, 4, torch.int64)])
def test_cache_matches_oracle_state_machine(oracle, dim, vpo, idx_t, async_wb):
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(dim)
    rows = [1000, 37, 5000, 64]          # unequal tables; the oracle takes one size: pad to the largest
    T = len(rows)
    n_slots, cap, ws = 700, 300, 8
    cache = EmbeddingCache(rows, dim, cache_size=cap, num_slots=n_slots, rows_per_page=vpo, evict_group=ws)
    cache.write(seed=3)
    cache.set_async_writeback(async_wb)
    page_bytes = cache.vpo * dim * 4
    pc = oracle.PageCache(T, max(rows), dim, cache_size=cap, page_bytes=page_bytes, ws_opt=ws, n_slots=n_slots)
    assert pc.vpo == cache.vpo
    expect = [s.copy() for s in cache.store]          # numpy model of every table's current contents
    for step in range(12):
        n_per = [int(x) for x in rng.integers(1, 120, size=T)]
        ids, offs = sorted_batch(rng, rows, n_per)
        d_ids = torch.from_numpy(ids).to(idx_t).cuda()
        got_rows = cache.get_emb_vectors(d_ids, offs)
        for t in range(T):
            pc.get(t, ids[offs[t]:offs[t + 1]])
        compare_state(cache, pc, T)
        # every id maps to slot*vpo + id%vpo and the pool row holds the table row
        r = got_rows.cpu().numpy().astype(np.int64)
        for t in range(T):
            seg = ids[offs[t]:offs[t + 1]]
            sm = pc.slot_map(t)
            assert np.array_equal(r[offs[t]:offs[t + 1]], sm[seg // cache.vpo] * cache.vpo + seg % cache.vpo)
            assert np.array_equal(cache.pool[got_rows[offs[t]:offs[t + 1]].long()].cpu().numpy(), expect[t][seg])
        # "train": bump the looked-up rows in the pool, mirror it in the numpy model
        uniq = torch.unique(got_rows.long())
        cache.pool[uniq] += 1.0
        for t in range(T):
            u = np.unique(ids[offs[t]:offs[t + 1]])
            expect[t][u] += 1.0
        cache.reclaim()
        pc.reclaim()
        compare_state(cache, pc, T)
    cache.flush()
    for t in range(T):
        assert np.array_equal(cache.store[t], expect[t]), "store of table %d after flush" % t
    cache.close()


def test_cache_reports_bad_id_and_exhausted_pool():
    from dlrm_amd.cache import EmbeddingCache
    cache = EmbeddingCache([100], 32, cache_size=4, num_slots=2, rows_per_page=1)
    cache.write()
    rows = cache.get_emb_vectors(torch.tensor([5, 100], dtype=torch.int64, device="cuda"), [0, 2])
    torch.cuda.synchronize()
    assert rows.cpu().tolist()[1] == -1 and cache.state().error_flags & 1
    cache.get_emb_vectors(torch.tensor([1, 2, 3], dtype=torch.int64, device="cuda"), [0, 3])
    assert cache.state().error_flags & 2          # 4 pages wanted, 2 slots
    cache.close()


def test_rmc1_reference_run_matches_reference_csv():
    """The reference's own RMC1 run (LRU_BASE_RMC1_C2_B40_D200.csv) replayed
    through the HBM cache: same ids (torch::manual_seed(0) dataset), same loop
    (src/dlrmtest.cpp:468-545), per-batch totals of pages read and written
    must equal the CSV's."""
    from dlrm_amd.cache import EmbeddingCache
    cfg = GOLD["rmc1"]
    T, D, L = cfg["tables"], cfg["dim"], cfg["lookups"]
    B, N, c, rows = cfg["batch"], cfg["data_size"], cfg["cache_mult"], cfg["rows_per_table"]
    _, ids, _ = reference_dataset(T, L, B, N, rows)
    cap = c * T * L * B
    cache = EmbeddingCache([rows] * T, D, cache_size=cap, num_slots=cap + T * L * B + 1024)
    trace = []
    for epoch in range(1, cfg["epochs"] + 1):
        for b, (s, bs) in enumerate(reference_batches(N, B)):
            per_table = [np.sort(ids[s:s + bs, t * L:(t + 1) * L].reshape(-1)) for t in range(T)]
            offs = np.concatenate([[0], np.cumsum([x.size for x in per_table])])
            cache.get_emb_vectors(torch.from_numpy(np.concatenate(per_table)).cuda(), offs)
            reads = cache.counters(reset=True)["read"]
            cache.reclaim()
            writes = cache.counters(reset=True)["write"]
            if b + 1 == c and epoch == 1:
                cache.cache_size = cache.page_count
            trace.append({"read": int(reads.sum()), "write": int(writes.sum())})
    assert cache.state().error_flags == 0
    assert trace == cfg["batches"]
    cache.close()


def test_store_image_roundtrip_through_the_cache(tmp_path):
    """An image in the reference's on-disk layout is loaded into the pinned store, rows are updated inside
    the HBM cache, and the saved image holds exactly the updated rows."""
    from dlrm_amd import ops, store_file
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(3)
    rows, D, T = [300, 77], 32, 2
    tables = [rng.random((r, D), dtype=np.float32) for r in rows]
    src, dst = tmp_path / "in.img", tmp_path / "out.img"
    store_file.write_store(src, tables)
    c = EmbeddingCache(rows, D, cache_size=8, num_slots=64)
    store_file.load_into(c, src)
    ids = torch.tensor([5, 200, 299, 0, 76, 5], dtype=torch.int32, device="cuda")     # 3 of table 0, 3 of table 1
    r = c.get_emb_vectors(ids, [0, 3, 6])
    got = c.pool[r.long()].cpu().numpy()
    want = np.stack([tables[0][5], tables[0][200], tables[0][299], tables[1][0], tables[1][76], tables[1][5]])
    assert np.array_equal(got, want)
    c.pool[r.long()[:3]] += 1.0                   # rows 5, 200, 299 of table 0 change inside the cache
    torch.cuda.synchronize()
    assert store_file.save_from(c, dst) == store_file.image_bytes(rows, D)
    back = store_file.read_store(dst, rows, D, copy=True)
    exp0 = tables[0].copy()
    exp0[[5, 200, 299]] += 1.0
    assert np.array_equal(back[0], exp0) and np.array_equal(back[1], tables[1])
    c.close()


def test_reclaim_of_one_table_matches_oracle(oracle):
    """cache::Reclaim_vector_page(table_ID) (include/data.h:267): the ws_opt least recently used pages of one table
    leave, page_count and that table's write counter move by ws_opt -- also when the table holds fewer pages."""
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(9)
    rows, dim, T = [800, 40, 3000], 32, 3
    cache = EmbeddingCache(rows, dim, cache_size=10 ** 6, num_slots=600, rows_per_page=4, evict_group=8)
    cache.write(seed=2)
    pc = oracle.PageCache(T, max(rows), dim, cache_size=10 ** 6, page_bytes=4 * dim * 4, ws_opt=8, n_slots=600)
    for step in range(10):
        ids, offs = sorted_batch(rng, rows, [int(x) for x in rng.integers(1, 60, size=T)])
        cache.get_emb_vectors(torch.from_numpy(ids).cuda(), offs)
        for t in range(T):
            pc.get(t, ids[offs[t]:offs[t + 1]])
        for t in ([step % T] if step % 2 else [1, 1, (step + 1) % T]):     # table 1 (10 pages) runs dry
            cache.reclaim_table(t)
            pc.reclaim_table(t)
            compare_state(cache, pc, T)
    assert cache.counters()["evicted"].sum() > 0
    cache.close()


@pytest.mark.parametrize("async_wb", [0, 10 ** 4], ids=["wb-in-reclaim", "wb-async"])
@pytest.mark.parametrize("vpo", [1, 8])
def test_dirty_only_writeback_matches_oracle_and_store(oracle, vpo, async_wb):
    """writeback_dirty_only: residency, eviction order and the reference's counters are unchanged; only pages
    marked dirty since they were read are copied back (counter 6), and the store still ends up with every update."""
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(21 + vpo)
    rows, dim = [2000, 500, 1500, 64], 16
    T = len(rows)
    n_slots, cap, ws = 1500, 300, 8
    cache = EmbeddingCache(rows, dim, cache_size=cap, num_slots=n_slots, rows_per_page=vpo, evict_group=ws,
                           dirty_only=True, max_lookup_ids=4 * 200)
    cache.write(seed=3)
    cache.set_async_writeback(async_wb)
    pc = oracle.PageCache(T, max(rows), dim, cache_size=cap, page_bytes=vpo * dim * 4, ws_opt=ws, n_slots=n_slots)
    pc.set_dirty_only(True)
    expect = [s.copy() for s in cache.store]
    for step in range(14):
        ids, offs = sorted_batch(rng, rows, [int(x) for x in rng.integers(1, 200, size=T)])
        got_rows = cache.get_emb_vectors(torch.from_numpy(ids).cuda(), offs)
        for t in range(T):
            pc.get(t, ids[offs[t]:offs[t + 1]])
        # tables 0 and 2 train (their rows change), 1 and 3 are only read
        for t in (0, 2):
            seg = got_rows[offs[t]:offs[t + 1]]
            cache.mark_dirty(seg)
            pc.mark_dirty(t, ids[offs[t]:offs[t + 1]])
            cache.pool[torch.unique(seg.long())] += 1.0
            expect[t][np.unique(ids[offs[t]:offs[t + 1]])] += 1.0
        cache.reclaim()
        pc.reclaim()
        compare_state(cache, pc, T)
        assert np.array_equal(cache.counters()["written_back"], pc.counters("written_back"))
    cnt = cache.counters()
    assert cnt["written_back"][[1, 3]].sum() == 0 and cnt["evicted"][[1, 3]].sum() > 0   # read-only tables cost nothing
    assert 0 < cnt["written_back"][[0, 2]].sum() <= cnt["evicted"][[0, 2]].sum()
    cache.flush()
    for t in range(T):
        assert np.array_equal(cache.store[t], expect[t]), "store of table %d" % t
    cache.close()


@pytest.mark.parametrize("async_wb", [0, 10 ** 4], ids=["wb-in-reclaim", "wb-async"])
@pytest.mark.parametrize("vpo", [1, 4])
@pytest.mark.parametrize("dirty_only", [False, True])
def test_row_state_travels_with_its_page(vpo, dirty_only, async_wb):
    """h_row_state_store: one float per row (an optimizer's state) is read into the pool's state array with the
    row's page and written back with it; exact bookkeeping of increments through many evictions."""
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(5 + vpo)
    rows, dim = [1000, 333, 64], 8
    T = len(rows)
    cache = EmbeddingCache(rows, dim, cache_size=120, num_slots=900, rows_per_page=vpo, evict_group=8,
                           dirty_only=dirty_only, max_lookup_ids=3 * 150, row_state=True)
    cache.write(seed=1)
    cache.set_async_writeback(async_wb)
    for t in range(T):
        cache.state_store[t][:] = np.arange(rows[t], dtype=np.float32) + 1000.0 * t
    expect = [s.copy() for s in cache.state_store]
    for step in range(12):
        ids, offs = sorted_batch(rng, rows, [int(x) for x in rng.integers(1, 150, size=T)])
        got_rows = cache.get_emb_vectors(torch.from_numpy(ids).cuda(), offs)
        torch.cuda.synchronize()
        st = cache.state_pool[got_rows.long()].cpu().numpy()
        for t in range(T):
            seg = ids[offs[t]:offs[t + 1]]
            assert np.array_equal(st[offs[t]:offs[t + 1]], expect[t][seg]), (step, t)
            expect[t][np.unique(seg)] += 1.0
        cache.mark_dirty(got_rows)
        cache.state_pool[torch.unique(got_rows.long())] += 1.0
        cache.reclaim()
    assert cache.counters()["evicted"].sum() > 0 and cache.state().error_flags & ~4 == 0
    cache.flush()
    torch.cuda.synchronize()
    for t in range(T):
        assert np.array_equal(cache.state_store[t], expect[t]), "state of table %d" % t
    cache.close()


@pytest.mark.parametrize("vpo", [1, 8])
def test_async_writeback_pages_wanted_again_at_once(vpo):
    """Every batch evicts what the batch before brought in and asks for much of it again: most misses of a lookup are
    pages still on their way to the store.  They must come back with the values they left with, batch after batch,
    and the store must end up with every update."""
    from dlrm_amd.cache import EmbeddingCache
    rng = np.random.default_rng(77 + vpo)
    rows, dim = [4096, 4096], 64
    n = 1500
    cache = EmbeddingCache(rows, dim, cache_size=16, num_slots=2 * 2 * n + 64, rows_per_page=vpo, evict_group=8)
    cache.write(seed=5)
    cache.set_async_writeback(2 * 2 * n + 64)
    expect = [s.copy() for s in cache.store]
    prev = None
    for step in range(10):
        per = []
        for t in range(2):
            fresh = rng.integers(0, rows[t], size=n // 3)
            again = prev[t][rng.integers(0, prev[t].size, size=n - n // 3)] if prev is not None else rng.integers(0, rows[t], size=n - n // 3)
            per.append(np.sort(np.concatenate([fresh, again])))
        prev = per
        ids = np.concatenate(per).astype(np.int64)
        offs = [0, n, 2 * n]
        got = cache.get_emb_vectors(torch.from_numpy(ids).cuda(), offs)
        for t in range(2):
            seg = ids[offs[t]:offs[t + 1]]
            assert np.array_equal(cache.pool[got[offs[t]:offs[t + 1]].long()].cpu().numpy(), expect[t][seg]), (step, t)
            expect[t][np.unique(seg)] += float(step + 1)
        cache.pool[torch.unique(got.long())] += float(step + 1)
        cache.reclaim()                 # capacity 16 pages: (nearly) everything leaves
    assert cache.state().error_flags & ~4 == 0 and cache.counters()["evicted"].sum() > 4 * n // vpo
    cache.flush()
    for t in range(2):
        assert np.array_equal(cache.store[t], expect[t]), "store of table %d" % t
    cache.close()
