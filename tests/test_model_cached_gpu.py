"""Training with the tables in pinned host DRAM behind the HBM page cache (DLRM(cache_fraction=...)):
every step looks its ids up, pools and updates rows inside the slot pool, reclaims; the next batch is
looked up on a second stream meanwhile.  Losses, predictions and -- after a flush -- the backing store
must match the oracle's step on whole tables (float32, 1e-5 relative)."""
import numpy as np
import pytest
import torch

from tests.oracle_model import make_batch, oracle_train_step
from tests.test_embbag_forward_gpu import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows_per_page", [1, 4])
@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("kind", ["dot", "cat"])
def test_cached_training_matches_oracle(oracle, rows_per_page, prefetch, kind):
    from dlrm_amd.model import DLRM, DLRMConfig
    if kind == "dot":
        cfg = DLRMConfig(table_rows=[600, 500, 1000, 640], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1], lr=0.05,
                         lookups=2)
    else:
        cfg = DLRMConfig(table_rows=[4000, 3000, 7000], dim=8, dense_in=6, bottom=[16, 8], top=[12, 1], interaction="cat",
                         loss="mse", final_relu=True, pooling="mean", lookups=3, lr=0.02)
    B = 32
    # capacities small enough that pages are evicted within the 8 steps, large enough that no page of
    # the running (or prefetched) batch is among the victims
    m = DLRM(cfg, B, seed=5, cache_fraction=0.5 if kind == "dot" else (0.1 if rows_per_page == 1 else 0.35),
             cache_rows_per_page=rows_per_page, replay=prefetch)   # the pipelined runs also replay a recorded step
    if kind == "cat":
        for b in m.bottom.b + m.top.b:
            b.abs_()
    T = cfg.num_tables
    tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
              for t, r in enumerate(cfg.table_rows)]
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    rng = np.random.default_rng(31)
    n_steps = 8
    host = [make_batch(rng, cfg, B) for _ in range(n_steps)]
    dev = [(torch.from_numpy(d).cuda(), torch.from_numpy(i.astype(np.int32)).cuda(),
            torch.from_numpy(o.astype(np.int32)).cuda(), torch.from_numpy(l).cuda()) for d, i, o, l in host]
    for s in range(n_steps):
        dense, idx, off, labels = host[s]
        d_dense, d_idx, d_off, d_labels = dev[s]
        m.load_batch(d_dense, d_labels)
        nxt = dev[s + 1][1] if (prefetch and s + 1 < n_steps) else None
        loss = m.train_step(d_idx, d_off, next_indices=nxt)
        torch.cuda.synchronize()
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        st = m.cache.state()
        assert st.error_flags == 0 and int(m.err_flag.item()) == 0
        if not prefetch:    # (pipelined: the second half of this step's reclaim is queued by the next step)
            assert st.page_count <= st.cache_size + T * 8      # reclaimed down to the capacity (+ one group per table)
    cnt = m.cache.counters()
    assert cnt["miss"].sum() > 0 and cnt["hit"].sum() > 0 and cnt["evicted"].sum() > 0
    m.cache.flush()
    for t in range(T):
        assert rel_err(np.asarray(m.cache.store[t]), tables[t]) <= 1e-5
    m.cache.close()


@pytest.mark.parametrize("rows_per_page", [1, 8])
def test_pipelined_training_with_a_cache_smaller_than_one_batch(oracle, rows_per_page):
    """The pipelined loop looks batch n+1 up before it reclaims for batch n.  With a capacity below one batch's
    pages every reclaim wants more victims than there are old pages; the pages of the batch just looked up must
    stay (protect age 1) or the next step would pool from recycled slots.  Zipf ids; losses, predictions and the
    flushed store against the oracle's step on whole tables."""
    from dlrm_amd.model import DLRM, DLRMConfig
    from tests.test_config_shapes_gpu import make_batch_dist
    cfg = DLRMConfig(table_rows=[20000, 16000, 24000, 20000], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1],
                     lr=0.05, lookups=4)
    B, T = 512, 4
    m = DLRM(cfg, B, seed=5, cache_fraction=0.02 if rows_per_page == 1 else 0.1, cache_rows_per_page=rows_per_page,
             replay=True)
    assert m.cache.state().cache_size < 2000          # one batch touches about 3300 / 2700 pages
    tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
              for t, r in enumerate(cfg.table_rows)]
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    rng = np.random.default_rng(41)
    n_steps = 8
    host = [make_batch_dist(rng, cfg, B, "zipf") for _ in range(n_steps)]
    dev = [(torch.from_numpy(d).cuda(), torch.from_numpy(i.astype(np.int32)).cuda(),
            torch.from_numpy(o.astype(np.int32)).cuda(), torch.from_numpy(l).cuda()) for d, i, o, l in host]
    for s in range(n_steps):
        dense, idx, off, labels = host[s]
        m.load_batch(dev[s][0], dev[s][3])
        loss = m.train_step(dev[s][1], dev[s][2], next_indices=dev[s + 1][1] if s + 1 < n_steps else None)
        torch.cuda.synchronize()
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        assert (m.cache.state().error_flags, int(m.err_flag.item())) == (0, 0), s
    assert m.cache.counters()["evicted"].sum() > 0
    m.cache.flush()
    for t in range(T):
        assert rel_err(np.asarray(m.cache.store[t]), tables[t]) <= 1e-5
    m.cache.close()


def test_inference_through_the_cache_writes_nothing_back(oracle):
    """DLRM.forward with cached tables: predictions of the oracle's forward pass; pages come and go but none is
    copied back to the store (nothing was updated)."""
    import copy
    from dlrm_amd.model import DLRM, DLRMConfig
    from tests.test_config_shapes_gpu import make_batch_dist
    cfg = DLRMConfig(table_rows=[20000, 16000, 24000, 20000], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1],
                     lr=0.05, lookups=4)
    B = 512
    m = DLRM(cfg, B, seed=5, cache_fraction=0.05, cache_rows_per_page=1)
    tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
              for t, r in enumerate(cfg.table_rows)]
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    before = [np.asarray(s).copy() for s in m.cache.store]
    rng = np.random.default_rng(43)
    for s in range(6):
        dense, idx, off, labels = make_batch_dist(rng, cfg, B, "zipf")
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        pred = m.forward(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda())
        torch.cuda.synchronize()
        _, pred_ref = oracle_train_step(oracle, cfg, copy.deepcopy(tables), copy.deepcopy(Ws), copy.deepcopy(bs), dense,
                                        idx, off, labels, len(cfg.bottom), cfg.lr)
        assert rel_err(pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        # (flag 4: the reclaim after a batch may drop that batch's own pages -- it is done with them)
        assert m.cache.state().error_flags & ~4 == 0
    cnt = m.cache.counters()
    assert cnt["evicted"].sum() > 0 and cnt["written_back"].sum() == 0
    m.cache.flush()
    for a, b in zip(before, m.cache.store):
        assert np.array_equal(a, np.asarray(b))
    m.cache.close()


@pytest.mark.parametrize("replay", [False, True], ids=["eager", "replay"])
@pytest.mark.parametrize("kind", ["dot", "cat"])
def test_training_through_the_vector_cache_matches_oracle(oracle, kind, replay):
    """DLRM(cache_policy="vec"): the tables sit behind the vector-granular cache (rows listed oldest request
    first, evicted rows repacked into new pages).  Losses and predictions must match the oracle's step on whole
    tables, and at the end every row must read back with the oracle's value wherever the cache keeps it."""
    from dlrm_amd.model import DLRM, DLRMConfig
    if kind == "dot":
        cfg = DLRMConfig(table_rows=[6000, 500, 9000, 640], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1], lr=0.05,
                         lookups=2)
    else:
        cfg = DLRMConfig(table_rows=[4000, 3000, 7000], dim=32, dense_in=6, bottom=[16, 32], top=[12, 1], interaction="cat",
                         loss="mse", final_relu=True, pooling="mean", lookups=3, lr=0.02)
    B = 128
    vpo = 4096 // (4 * cfg.dim)
    frac = 8 * vpo / max(cfg.table_rows) * 1.05          # lists barely longer than one write buffer: evictions from step 2 on
    m = DLRM(cfg, B, seed=5, cache_fraction=frac, cache_policy="vec", replay=replay)
    if kind == "cat":
        for b in m.bottom.b + m.top.b:
            b.abs_()
    T = cfg.num_tables
    tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
              for t, r in enumerate(cfg.table_rows)]
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    rng = np.random.default_rng(33)
    for s in range(8):
        dense, idx, off, labels = make_batch(rng, cfg, B)
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        loss = m.train_step(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda())
        torch.cuda.synchronize()
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        assert m.cache.state().error_flags == 0 and int(m.err_flag.item()) == 0
    cnt = m.cache.counters()
    assert cnt["miss"].sum() > 0 and cnt["hit"].sum() > 0 and cnt["evicted"].sum() > 0 and cnt["write"].sum() > 0
    for t in range(T):
        ids = torch.arange(cfg.table_rows[t], dtype=torch.int32, device="cuda")
        offs = [0] * (t + 1) + [cfg.table_rows[t]] * (T - t)
        assert rel_err(m.cache.read_rows(ids, offs).cpu().numpy(), tables[t]) <= 1e-5
    m.cache.close()


@pytest.mark.parametrize("policy,rows_per_page", [("lru", 1), ("lru", 4), ("vec", None)])
@pytest.mark.parametrize("rows_t,emb_opt", [("f32", "adagrad"), ("bf16", "sgd"), ("bf16", "adagrad")])
def test_cached_training_with_adagrad_state_and_bf16_rows(oracle, policy, rows_per_page, rows_t, emb_opt):
    """Row-wise Adagrad and bf16 rows through both caches: a row's state is read, kept and written back with the row
    (page-LRU: with its page; vector cache: store <-> write buffer <-> pool), so training with most rows outside HBM
    gives the oracle's step on whole tables -- losses, predictions, and at the end every row and every state."""
    from dlrm_amd.model import DLRM, DLRMConfig
    from tests.test_config_shapes_gpu import make_batch_dist
    from tests.test_model_gpu import bf16_rows_close
    # (mean pooling and a small step keep this little model well conditioned over the 8 steps)
    cfg = DLRMConfig(table_rows=[3000, 2500, 4000, 3500], dim=32, dense_in=13, bottom=[32, 32], top=[24, 1], lr=0.01,
                     lookups=3, emb_optimizer=emb_opt, row_dtype=rows_t, pooling="mean")
    B, T = 128, 4
    kw = dict(cache_policy="vec", cache_fraction=0.12, cache_log_pages=512) if policy == "vec" else \
        dict(cache_fraction=0.15 if rows_per_page == 1 else 0.4, cache_rows_per_page=rows_per_page)
    m = DLRM(cfg, B, seed=5, **kw)
    if rows_t == "bf16":
        tables = [np.asarray(s).copy() for s in m.cache.store]          # uint16: the bf16 bit patterns
    else:
        tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
                  for t, r in enumerate(cfg.table_rows)]
    tro = np.concatenate([[0], np.cumsum(cfg.table_rows)])
    state = np.zeros(int(tro[-1]), dtype=np.float32) if emb_opt == "adagrad" else None
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    rng = np.random.default_rng(47)
    for s in range(8):
        dense, idx, off, labels = make_batch_dist(rng, cfg, B, "zipf")
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        loss = m.train_step(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda())
        torch.cuda.synchronize()
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr, emb_opt_state=state)
        # bf16 rows: a row one bf16 step off moves a pooled vector by up to 2^-9 of one addend
        tol = 1e-5 if rows_t == "f32" else 2e-3
        assert abs(loss.item() - loss_ref) <= tol * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= tol
        assert (m.cache.state().error_flags, int(m.err_flag.item())) == (0, 0), s
    assert m.cache.counters()["evicted"].sum() > 0
    # every row and every state, wherever it lives now
    for t, r in enumerate(cfg.table_rows):
        if policy == "vec":
            ids = torch.arange(r, dtype=torch.int32, device="cuda")
            offs = [0] * (t + 1) + [r] * (T - t)
            got, got_state = m.cache.read_rows(ids, offs, with_state=True) if state is not None else \
                (m.cache.read_rows(ids, offs), None)
            got = got.view(torch.int16).cpu().numpy().view(np.uint16) if rows_t == "bf16" else got.cpu().numpy()
            got_state = None if got_state is None else got_state.cpu().numpy()
        else:
            if t == 0:
                m.cache.flush()
                torch.cuda.synchronize()
            got = np.asarray(m.cache.store[t])
            got_state = None if state is None else np.asarray(m.cache.state_store[t])
        if rows_t == "bf16":
            got_f, ref_f = oracle.bf16_to_f32(got), oracle.bf16_to_f32(tables[t])
            assert np.all(np.abs(got_f - ref_f) <= 2.0 ** -6 * np.abs(ref_f) + 1e-30)      # up to a step per update
            assert np.mean(got == tables[t]) >= 0.99
        else:
            assert rel_err(got, tables[t]) <= 1e-5
        if state is not None:
            assert rel_err(got_state, state[tro[t]:tro[t + 1]]) <= (1e-5 if rows_t == "f32" else 2e-3)
    m.cache.close()
