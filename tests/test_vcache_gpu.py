"""The vector-granular HBM cache (dlrm_vcache_*, the reference's class cache2) against the CPU oracle: after
every call the state machine must agree bit for bit -- per table the pages read / written and the ids found
resident or not, per row residency, list membership, hit count and page, the order of the list and of the
write buffer, page_count -- and every row must read back with its latest value wherever it lives.  The replay
of the reference's own dlrmtest2 run is checked against the counters in its CSV
(tests/golden/reference_lru_vec_trace.json)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_vec_trace.json")))


class Pair:
    """The HIP cache and the oracle fed with the same calls."""

    def __init__(self, oracle, T, rows, dim, cache_size, num_slots, log_pages, idx_t=torch.int64):
        from dlrm_amd.vcache import VectorCache
        self.T, self.rows, self.dim, self.idx_t = T, rows, dim, idx_t
        self.gpu = VectorCache([rows] * T, dim, cache_size, num_slots, log_pages)
        self.cpu = oracle.VectorCache(T, rows, dim, cache_size)
        self.gpu.write(seed=3)
        self.truth = [s.copy() for s in self.gpu.store]          # latest value of every row

    def _flat(self, req):
        offs = np.concatenate([[0], np.cumsum([len(r) for r in req])])
        ids = torch.from_numpy(np.concatenate(req).astype(np.int64)).to(self.idx_t).cuda()
        return ids, offs

    def lookup(self, req, update=True):
        ids, offs = self._flat(req)
        rows = self.gpu.get_emb_vectors(ids, offs)
        for t in range(self.T):
            self.cpu.get(t, req[t])
        assert int(rows.min()) >= 0
        if update:       # what the fused optimizer does: the asked rows change inside the pool
            for t in range(self.T):
                u = np.unique(req[t])
                self.truth[t][u] += 1.0 + t
            uniq = torch.unique(rows.long())
            tab = torch.zeros(self.gpu.num_slots, device="cuda")
            for t in range(self.T):
                tab[rows[offs[t]:offs[t + 1]].long()] = 1.0 + t
            self.gpu.pool[uniq] += tab[uniq, None]
        return rows

    def reclaim(self):
        self.gpu.reclaim()
        self.cpu.reclaim()

    def restore(self, req):
        ids, offs = self._flat(req)
        self.gpu.restore(ids, offs)
        for t in range(self.T):
            self.cpu.save(t, req[t])

    def check(self, values=True):
        st = self.gpu.state()
        assert st.error_flags == 0
        assert st.page_count == self.cpu.page_count
        c = self.gpu.counters()
        for name in ("read", "write", "hit", "miss"):
            assert c[name].tolist() == self.cpu.counters(name).tolist(), name
        for t in range(self.T):
            res, lst, cnt, page, order, pend = self.gpu.table_view(t)
            ores, olst, ocnt, oloc = self.cpu.export(t)
            assert np.array_equal(res, ores.astype(bool)), "resident rows of table %d" % t
            assert np.array_equal(lst, olst.astype(bool)), "listed rows of table %d" % t
            assert np.array_equal(cnt[lst], ocnt[lst]), "hit counts of table %d" % t
            assert np.array_equal(page, oloc), "pages of table %d" % t
            assert order.tolist() == self.cpu.fifo_order(t).tolist(), "list order of table %d" % t
            waiting = np.nonzero(pend >= 0)[0]
            assert waiting[np.argsort(pend[waiting])].tolist() == self.cpu.pending_rows(t).tolist()
            assert c["listed"][t] == lst.sum() and c["resident"][t] == res.sum()
            assert c["pending"][t] == self.cpu.pending(t) and c["written"][t] == self.cpu.written_pages(t)
        if values:
            for t in range(self.T):
                ids = torch.arange(self.rows, dtype=self.idx_t, device="cuda")
                offs = [0] * (t + 1) + [self.rows] * (self.T - t)
                got = self.gpu.read_rows(ids, offs).cpu().numpy()
                assert np.array_equal(got, self.truth[t]), "values of table %d" % t

    def close(self):
        self.gpu.close()


@pytest.mark.parametrize("idx_t", [torch.int64, torch.int32])
@pytest.mark.parametrize("dim,zipf", [(32, False), (64, True), (128, False)])
def test_state_machine_matches_oracle(oracle, dim, zipf, idx_t):
    T, rows = 3, 6000
    vpo = 4096 // (4 * dim)
    W = 8 * vpo
    p = Pair(oracle, T, rows, dim, cache_size=max(W, 300), num_slots=T * rows, log_pages=2048, idx_t=idx_t)
    rng = np.random.default_rng(dim)
    try:
        p.check()
        for step in range(14):
            if zipf:
                req = [np.minimum(rng.zipf(1.2, 700) - 1, rows - 1).astype(np.int64) for _ in range(T)]
                req = [(r * 37 + 11 * t) % rows for t, r in enumerate(req)]
            else:
                req = [rng.integers(0, rows, 500 + 100 * t) for t in range(T)]
            if step % 5 == 4:
                req[1] = req[1][:0]                      # a table without lookups in this batch
            p.lookup(req)
            p.check(values=False)
            p.reclaim()
            p.check(values=False)
            p.restore(req)
            p.check(values=step % 4 == 3)
        c = p.gpu.counters()
        assert c["write"].sum() > 0 and c["dropped"].sum() > 0 and c["evicted"].sum() > 0
    finally:
        p.close()


def test_reclaim_spanning_several_buffers_and_repeated_rows(oracle):
    # one call evicts more than two buffers' worth; lookups repeat rows inside one call and re-read rows that
    # still wait in the write buffer
    T, rows, dim = 2, 4000, 32
    p = Pair(oracle, T, rows, dim, cache_size=256, num_slots=T * rows, log_pages=1024)
    rng = np.random.default_rng(5)
    try:
        first = [np.concatenate([np.arange(0, 1500), np.arange(0, 1500, 7)]) for _ in range(T)]
        p.lookup(first)
        p.check(values=False)
        p.reclaim()                                        # 1244 evictions per table: 4 buffers + 220 waiting
        p.check()
        again = [np.concatenate([np.arange(1000, 1300), rng.integers(0, rows, 200)]) for _ in range(T)]
        p.lookup(again)                                    # rows 1024..1243 wait in the buffer: values must be current
        p.check()
        p.reclaim()
        p.restore(again)
        p.check()
    finally:
        p.close()


def test_errors(oracle):
    from dlrm_amd.vcache import VectorCache
    from dlrm_amd._capi import DlrmError
    with pytest.raises(DlrmError, match="cache_size"):
        VectorCache([1000], 32, cache_size=10, num_slots=100, log_pages_per_table=64)
    vc = VectorCache([1000], 32, cache_size=256, num_slots=40, log_pages_per_table=64)
    try:
        ids = torch.arange(0, 1000, 32, device="cuda")               # 32 pages x 32 rows > 40 slots
        vc.get_emb_vectors(ids, [0, ids.numel()])
        assert vc.state().error_flags & 2
        bad = torch.tensor([5, 1000], device="cuda")
        vc.get_emb_vectors(bad, [0, 2])
        assert vc.state().error_flags & 1
    finally:
        vc.close()


@pytest.mark.parametrize("name,slots,log_pages", [("vec_t25_d32_l20", 4_500_000, 512),
                                                  ("vec_t32_d64_l120", 10_000_000, 4096)])
def test_replays_reference_dlrmtest2_run(oracle, name, slots, log_pages):
    """The reference's own runs (25 tables x 800000 rows x 32 with 20 lookups; 32 x 500000 x 64 with 120
    lookups; batch 40, 10 batches): per batch and table the pages read and written that its CSVs record."""
    from dlrm_amd.vcache import VectorCache
    from tests.test_oracle_veccache_csv import run_trace
    cfg = GOLD[name]

    class Replay:
        def __init__(self, T, rows, D, cache_size):
            self.T = T
            self.vc = VectorCache([rows] * T, D, cache_size, num_slots=slots, log_pages_per_table=log_pages)

        def _one(self, t, ids):
            d = torch.from_numpy(np.ascontiguousarray(ids)).cuda()
            return d, [0] * (t + 1) + [d.numel()] * (self.T - t)

        def get(self, t, ids):
            self.vc.get_emb_vectors(*self._one(t, ids))

        def save(self, t, ids):
            self.vc.restore(*self._one(t, ids))

        def reclaim(self):
            self.vc.reclaim()

        def counters(self, which, reset=False):
            return self.vc.counters(reset=reset)[which]

    holder = {}

    def make(T, rows, D, cs):
        holder["r"] = Replay(T, rows, D, cs)
        return holder["r"]
    try:
        got = run_trace(make, cfg)
        assert holder["r"].vc.state().error_flags == 0
        for b, (g, w) in enumerate(zip(got, cfg["batches"])):
            assert g == w, "batch %d" % b
    finally:
        holder["r"].vc.close()


@pytest.mark.parametrize("rows_t", [torch.float32, torch.bfloat16])
def test_row_values_and_state_follow_the_row(rows_t):
    """bf16 rows (64 rows per 4096-byte page at dim 32) and one float of optimizer state per row: both follow a row
    from the store into the pool, into the write buffer on eviction, into a newly written page, and back."""
    from dlrm_amd.vcache import VectorCache
    T, rows, dim = 2, 3000, 32
    vpo = 4096 // (dim * (4 if rows_t == torch.float32 else 2))
    gpu = VectorCache([rows] * T, dim, cache_size=8 * vpo, num_slots=T * rows, log_pages_per_table=1024, row_dtype=rows_t,
                      row_state=True)
    try:
        val = [np.zeros(rows, dtype=np.float32) for _ in range(T)]       # every row holds one small integer (exact in bf16)
        state = [np.arange(rows, dtype=np.float32) * 0.5 + 7.0 * t for t in range(T)]
        for t in range(T):
            if rows_t == torch.float32:
                gpu.store[t][:] = 0.0
            else:
                gpu.store[t][:] = 0
            gpu.state_store[t][:rows] = state[t]
        rng = np.random.default_rng(3)
        for step in range(10):
            req = [rng.integers(0, rows, 400) for _ in range(T)]
            ids = torch.from_numpy(np.concatenate(req).astype(np.int32)).cuda()
            offs = [0, len(req[0]), len(req[0]) + len(req[1])]
            got = gpu.get_emb_vectors(ids, offs)
            torch.cuda.synchronize()
            assert int(got.min()) >= 0
            exp_state = np.concatenate([state[t][req[t]] for t in range(T)])
            exp_val = np.concatenate([val[t][req[t]] for t in range(T)])
            assert np.array_equal(gpu.state_pool[got.long()].cpu().numpy(), exp_state), step
            assert np.array_equal(gpu.pool[got.long(), 0].float().cpu().numpy(), exp_val), step
            uniq = torch.unique(got.long())
            gpu.pool[uniq] += 1.0
            gpu.state_pool[uniq] += 0.25
            for t in range(T):
                u = np.unique(req[t])
                val[t][u] += 1.0
                state[t][u] += 0.25
            gpu.reclaim()
            gpu.restore(ids, offs)
        c = gpu.counters()
        assert gpu.state().error_flags == 0 and c["write"].sum() > 0 and c["evicted"].sum() > 0
        for t in range(T):
            ids = torch.arange(rows, dtype=torch.int32, device="cuda")
            offs = [0] * (t + 1) + [rows] * (T - t)
            got, got_state = gpu.read_rows(ids, offs, with_state=True)
            assert np.array_equal(got.float().cpu().numpy(), np.repeat(val[t][:, None], dim, axis=1)), t
            assert np.array_equal(got_state.cpu().numpy(), state[t]), t
    finally:
        gpu.close()
