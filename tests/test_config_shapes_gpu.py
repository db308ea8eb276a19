"""The benchmark configurations at their real shapes against the CPU oracle.

Criteo-Kaggle: 26 tables with the Kaggle cardinalities, dim 16, batch 2048, bottom 13-512-256-64-16,
dot interaction (27 features -> 351 pairs), top 367-512-256-1, BCE -- exactly DLRMConfig.kaggle(); only the
row counts are capped so that the oracle's copy of the tables stays small (no kernel sees a table's
row count as a shape: the EmbeddingBag kernels address rows through offsets).  Uniform and Zipf(1.05) ids,
eager and replayed, float32 and bfloat16 rows, SGD and row-wise Adagrad.

Tolerances: loss 1e-5 relative; predictions, activations and dense parameters 1e-5 of the tensor's largest
|reference| value; bfloat16 rows 1e-2; float32 rows element by element |got - ref| <= 1e-5 max|ref| +
32 * 2^-24 * step * sum_occurrences |g|, where step = lr (SGD) or lr / (sqrt(state) + eps) (Adagrad): a row hit by
hundreds of samples receives a float32 sum of gradients that largely cancel, and Adagrad's first steps divide
that sum by its own magnitude, so the rounding of the accumulation (float32 on the device as in libtorch's
index_add_, double in the oracle) is measured against the absolute sum, not the cancelled one.

The backward pass of the oracle is evaluated at the device's ReLU pattern (tests/oracle_model.py explains why);
the device's activations are compared with the oracle's own, and the units on which the two patterns differ
are counted (a few per million)."""
import numpy as np
import pytest
import torch

from tests.oracle_model import oracle_train_step
from tests.test_embbag_forward_gpu import rel_err

pytestmark = pytest.mark.gpu


def zipf_ids(rng, rows, n, alpha=1.05):
    """Exact Zipf(alpha) over ranks 1..rows (inverse CDF), ranks scattered over the table."""
    p = 1.0 / np.arange(1, rows + 1, dtype=np.float64) ** alpha
    cdf = np.cumsum(p)
    r = np.searchsorted(cdf, rng.random(n) * cdf[-1])
    return ((r.astype(np.int64) * 2654435761 + 12345) % rows).astype(np.int64)


def make_batch_dist(rng, cfg, B, dist):
    T, L = cfg.num_tables, cfg.lookups
    off = (np.arange(T * B + 1, dtype=np.int64) * L)
    idx = np.empty(T * B * L, dtype=np.int64)
    for t in range(T):
        n = B * L
        idx[t * n:(t + 1) * n] = zipf_ids(rng, cfg.table_rows[t], n) if dist == "zipf" else \
            rng.integers(0, cfg.table_rows[t], size=n)
    dense = rng.standard_normal((B, cfg.dense_in)).astype(np.float32)
    labels = (rng.random(B) < 0.3).astype(np.float32) if cfg.loss == "bce" else rng.standard_normal(B).astype(np.float32)
    return dense, idx, off, labels


def kaggle_cfg(cap=30000, **kw):
    from dlrm_amd.model import DLRMConfig
    cfg = DLRMConfig.kaggle(lr=0.1, **kw)
    cfg.table_rows = [min(r, cap) for r in cfg.table_rows]
    return cfg


EPS32 = 2.0 ** -24


def row_tolerance(cfg, t, idx, off, B, g_pooled, ref_rows, step_scale):
    """Per element tolerance of table t's updated rows (see the module docstring)."""
    lo, hi = int(off[t * B]), int(off[(t + 1) * B])
    ids = idx[lo:hi]
    lens = np.diff(off[t * B:(t + 1) * B + 1])
    bags = np.repeat(np.arange(B), lens)
    g = np.abs(g_pooled[bags, t, :]).astype(np.float64)
    if cfg.pooling == "mean":
        g /= np.maximum(lens[bags], 1)[:, None]
    A = np.zeros(ref_rows.shape, dtype=np.float64)
    np.add.at(A, ids, g)
    return 1e-5 * np.abs(ref_rows).max() + 32 * EPS32 * step_scale * A


def state_of(m):
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    tro = m.table_row_offsets.cpu().numpy()
    tables = [m.weight[tro[i]:tro[i + 1]].float().cpu().numpy().copy() for i in range(m.Tl)]
    return tables, Ws, bs


@pytest.mark.parametrize("replay", [False, True], ids=["eager", "replay"])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
@pytest.mark.parametrize("rows,opt", [("f32", "sgd"), ("f32", "adagrad"), ("bf16", "sgd"), ("bf16", "adagrad")])
def test_kaggle_shape_train_step_matches_oracle(oracle, replay, dist, rows, opt):
    from dlrm_amd.model import DLRM
    cfg = kaggle_cfg(row_dtype=rows, emb_optimizer=opt)
    assert cfg.num_tables == 26 and cfg.dim == 16 and cfg.top_in == 367
    assert [cfg.dense_in] + cfg.bottom == [13, 512, 256, 64, 16] and cfg.top == [512, 256, 1]
    B = 2048
    m = DLRM(cfg, B, seed=11, replay=replay)
    tables, Ws, bs = state_of(m)
    state = None if m.opt_state is None else m.opt_state.cpu().numpy().copy()
    row_tol = 1e-5 if rows == "f32" else 1e-2
    rng = np.random.default_rng(5)
    for step in range(3):
        dense, idx, off, labels = make_batch_dist(rng, cfg, B, dist)
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        loss = m.train_step(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda())
        torch.cuda.synchronize()
        assert int(m.err_flag.item()) == 0
        pattern = [a.cpu().numpy() for a in m.bot_acts + m.top_acts]
        extras = {}
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr, emb_opt_state=state, relu_pattern=pattern, extras=extras)
        flips = units = 0
        for got, ref in zip(pattern[:-1], extras["acts"][:-1]):      # every ReLU layer (the logit has none)
            assert rel_err(got, ref) <= 1e-5
            flips += int(((got > 0) != (ref > 0)).sum())
            units += got.size
        assert flips <= 2e-5 * units, (flips, units)
        if rows == "bf16":
            # the oracle keeps float32 rows; the device rounds every updated row to bfloat16 -- carry the
            # device's rows forward so that the comparison stays a per-step one
            got_t, _, _ = state_of(m)
            for a, r in zip(got_t, tables):
                assert rel_err(a, r) <= row_tol
                r[...] = a
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        got_t, got_W, got_b = state_of(m)
        tro = m.table_row_offsets.cpu().numpy()
        for t, (a, r) in enumerate(zip(got_t, tables)):
            if rows == "bf16":
                assert rel_err(a, r) <= row_tol
                continue
            scale = cfg.lr if state is None else (cfg.lr / (np.sqrt(state[tro[t]:tro[t + 1]]) + cfg.eps))[:, None]
            tol = row_tolerance(cfg, t, idx, off, B, extras["g_pooled"], r, scale)
            assert (np.abs(a.astype(np.float64) - r) <= tol).all(), (step, t, float((np.abs(a - r) / tol).max()))
        errs = {("W%d" % i if i < len(Ws) else "b%d" % (i - len(Ws))): rel_err(a, r)
                for i, (a, r) in enumerate(zip(got_W + got_b, Ws + bs))}
        assert max(errs.values()) <= 1e-5, (step, errs)
        if state is not None:
            assert rel_err(m.opt_state.cpu().numpy(), state) <= 1e-5


@pytest.mark.parametrize("rows_per_page", [1, 8])
@pytest.mark.parametrize("prefetch", [False, True], ids=["inline", "prefetch"])
def test_cached_training_on_zipf_ids_matches_oracle(oracle, rows_per_page, prefetch):
    """Tables behind the HBM page cache, Zipf(1.05) ids: a few hot rows appear hundreds of times per batch
    (duplicate-heavy miss claims and counter updates).  Losses, predictions, and the flushed store must match
    the oracle's step on whole tables; the hit / miss / read counters must match the oracle's page cache driven
    with the same (sorted per table, as the reference's driver passes them) ids."""
    from dlrm_amd.model import DLRM, DLRMConfig
    cfg = DLRMConfig(table_rows=[20000, 16000, 24000, 20000], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1],
                     lr=0.05, lookups=4)
    B = 512
    T = cfg.num_tables
    # capacities that hold two batches' pages (this one's and the prefetched one's: about 3300 / 2700 each) but
    # not what eight batches touch, so pages are evicted from the third step on
    m = DLRM(cfg, B, seed=5, cache_fraction=0.1 if rows_per_page == 1 else 0.6, cache_rows_per_page=rows_per_page,
             replay=prefetch)
    tables = [oracle.table_init_uniform(r * cfg.dim, (5 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
              for t, r in enumerate(cfg.table_rows)]
    Ws = [w.cpu().numpy().copy() for w in m.bottom.W + m.top.W]
    bs = [b.cpu().numpy().copy() for b in m.bottom.b + m.top.b]
    st0 = m.cache.state()
    pc = oracle.PageCache(T, max(cfg.table_rows), cfg.dim, cache_size=st0.cache_size,
                          page_bytes=rows_per_page * cfg.dim * 4, ws_opt=8, n_slots=st0.num_slots)
    rng = np.random.default_rng(77)
    n_steps = 8
    host = [make_batch_dist(rng, cfg, B, "zipf") for _ in range(n_steps)]
    dev = [(torch.from_numpy(d).cuda(), torch.from_numpy(i.astype(np.int32)).cuda(),
            torch.from_numpy(o.astype(np.int32)).cuda(), torch.from_numpy(l).cuda()) for d, i, o, l in host]
    dup = 0
    for s in range(n_steps):
        dense, idx, off, labels = host[s]
        dup = max(dup, int(np.bincount(idx[:B * cfg.lookups]).max()))
        m.load_batch(dev[s][0], dev[s][3])
        nxt = dev[s + 1][1] if (prefetch and s + 1 < n_steps) else None
        loss = m.train_step(dev[s][1], dev[s][2], next_indices=nxt)
        torch.cuda.synchronize()
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                               cfg.lr)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        assert (m.cache.state().error_flags, int(m.err_flag.item())) == (0, 0), s
        if not prefetch:
            # the state machine, call by call (the pipelined order looks the next batch up before this
            # step's reclaim, a different -- also deterministic -- sequence of calls)
            n = B * cfg.lookups
            for t in range(T):
                pc.get(t, np.sort(idx[t * n:(t + 1) * n]))
            pc.reclaim()
            cnt = m.cache.counters()
            for name in ("read", "write", "hit", "miss"):
                assert np.array_equal(cnt[name], pc.counters(name)), name
            assert m.cache.state().page_count == pc.page_count
            for t in range(T):
                got = m.cache.page_map(t).astype(np.int64)
                assert np.array_equal(got, pc.slot_map(t)[:got.size])
    assert dup >= 20          # the batches really are duplicate-heavy
    cnt = m.cache.counters()
    assert cnt["miss"].sum() > 0 and cnt["hit"].sum() > 0 and cnt["evicted"].sum() > 0
    m.cache.flush()
    for t in range(T):
        assert rel_err(np.asarray(m.cache.store[t]), tables[t]) <= 1e-5
    m.cache.close()
