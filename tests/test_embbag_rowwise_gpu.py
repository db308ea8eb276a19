"""Row-wise sharded tables: every shard holds one contiguous row range of each table and is
handed the GLOBAL indices (dlrm_embbag.d_shard_row_begin).  The shards' pooled outputs must add
up to the oracle's output on the whole tables, and the shards' rows after the fused update must
be the oracle's updated rows.  Tolerances as in the unsharded tests (float32 1e-5 relative;
a bag whose rows all sit in one shard is bit-exact)."""
import numpy as np
import pytest
import torch

from tests.test_embbag_forward_gpu import make_case, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # T, B, D, rows, L, ragged, shards
    (1, 33, 128, [1000], 1, False, 2),
    (3, 40, 64, [300, 8, 77], 5, True, 3),
    (2, 17, 32, [500, 20], 80, False, 4),        # long bags: wave-per-bag kernel inside a shard
    (2, 9, 20, [11, 40], 4, True, 2),            # element-per-lane path
]


def shard_tables(w, tro, shards):
    """Per shard: (weight rows, local table_row_offsets, first global row per table)."""
    T = len(tro) - 1
    out = []
    for s in range(shards):
        parts, begin, ltro = [], [], [0]
        for t in range(T):
            rows = int(tro[t + 1] - tro[t])
            lo, hi = rows * s // shards, rows * (s + 1) // shards
            parts.append(w[tro[t] + lo:tro[t] + hi])
            begin.append(lo)
            ltro.append(ltro[-1] + hi - lo)
        out.append((np.concatenate(parts), np.asarray(ltro, np.int64), np.asarray(begin, np.int64)))
    return out


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("mode", ["sum", "mean"])
def test_shard_outputs_add_up(oracle, case, idx_t, mode):
    from dlrm_amd import ops
    T, B, D, rows_list, L, ragged, shards = case
    rng = np.random.default_rng(T * 1000 + B)
    w, tro, idx, off = make_case(rng, T, B, D, rows_list, L, ragged)
    pooling = oracle.POOL_MEAN if mode == "mean" else oracle.POOL_SUM
    ref = oracle.embbag_forward(w, idx, off, B, T, tro, pooling, True)
    d_idx, d_off = torch.from_numpy(idx).to(idx_t).cuda(), torch.from_numpy(off).to(idx_t).cuda()
    total = np.zeros_like(ref)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for ws, ltro, begin in shard_tables(w, tro, shards):
        out = ops.embbag_forward(torch.from_numpy(np.ascontiguousarray(ws)).cuda(), d_idx, d_off, B, T,
                                 torch.from_numpy(ltro).cuda(), pooling, True, error_flag=flag,
                                 shard_row_begin=torch.from_numpy(begin).cuda())
        total += out.cpu().numpy()
    assert flag.item() == 0                       # rows of other shards are not errors
    assert rel_err(total, ref) <= 1e-5
    if L == 1 and not ragged:
        assert np.array_equal(total, ref)         # one row per bag: exactly one shard contributes


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_shard_updates_match_whole_table_update(oracle, case, mode, opt):
    from dlrm_amd import ops
    T, B, D, rows_list, L, ragged, shards = case
    rng = np.random.default_rng(T * 77 + B)
    w, tro, idx, off = make_case(rng, T, B, D, rows_list, L, ragged)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    pooling = oracle.POOL_MEAN if mode == "mean" else oracle.POOL_SUM
    optimizer = oracle.OPT_ADAGRAD_ROWWISE if opt == "adagrad" else oracle.OPT_SGD
    state = (rng.random(w.shape[0]).astype(np.float32) * 0.1) if opt == "adagrad" else None
    w_ref = w.copy()
    st_ref = None if state is None else state.copy()
    oracle.embbag_backward_update(w_ref, idx, off, B, grad, T, tro, pooling, True, optimizer, oracle.DUP_ACCUMULATE,
                                  0.05, 1e-8, st_ref)
    d_idx, d_off = torch.from_numpy(idx).cuda(), torch.from_numpy(off).cuda()
    d_grad = torch.from_numpy(grad).cuda()
    scale = max(float(np.max(np.abs(w_ref))), 1e-30)
    for s, (ws, ltro, begin) in enumerate(shard_tables(w, tro, shards)):
        dw = torch.from_numpy(np.ascontiguousarray(ws)).cuda()
        d_state = None
        if state is not None:
            st_parts = [state[tro[t] + begin[t]:tro[t] + begin[t] + (ltro[t + 1] - ltro[t])] for t in range(T)]
            d_state = torch.from_numpy(np.concatenate(st_parts)).cuda()
        ops.embbag_backward_update(dw, d_idx, d_off, B, d_grad, T, torch.from_numpy(ltro).cuda(), pooling, True,
                                   optimizer=optimizer, lr=0.05, eps=1e-8, opt_state=d_state,
                                   shard_row_begin=torch.from_numpy(begin).cuda())
        got = dw.cpu().numpy()
        for t in range(T):
            n = int(ltro[t + 1] - ltro[t])
            want = w_ref[tro[t] + begin[t]:tro[t] + begin[t] + n]
            assert np.max(np.abs(got[ltro[t]:ltro[t] + n] - want)) <= 1e-5 * scale
            if d_state is not None:
                want_s = st_ref[tro[t] + begin[t]:tro[t] + begin[t] + n]
                assert np.allclose(d_state.cpu().numpy()[ltro[t]:ltro[t] + n], want_s, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("halves", [False, True], ids=["update", "prepare+apply"])
def test_descriptor_for_some_tables_of_a_longer_index_array(oracle, split, halves):
    """Two descriptors share one index / offset array (tables [0, split) and [split, T), the offsets keep
    their positions in the whole array): the fused update of each must ignore the positions that belong
    to the other, whatever an earlier call left in the workspace."""
    from dlrm_amd import ops
    T, B, D, rows_list, L = 3, 37, 32, [300, 8, 77], 4
    rng = np.random.default_rng(91 + split)
    w, tro, idx, off = make_case(rng, T, B, D, rows_list, L, True)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    w_ref = w.copy()
    oracle.embbag_backward_update(w_ref, idx, off, B, grad, T, tro, oracle.POOL_SUM, True, oracle.OPT_SGD,
                                  oracle.DUP_ACCUMULATE, 0.05, 1e-8, None)
    dw = torch.from_numpy(w).cuda()
    d_idx, d_off = torch.from_numpy(idx).cuda(), torch.from_numpy(off).cuda()
    d_tro = torch.from_numpy(tro).cuda()
    d_grad = torch.from_numpy(grad).cuda()
    for t0, t1 in ((0, split), (split, T)):
        kw = dict(weight=dw, indices=d_idx, offsets=d_off[t0 * B:t1 * B + 1], batch=B, num_tables=t1 - t0,
                  table_row_offsets=d_tro[t0:t1 + 1], pooling=oracle.POOL_SUM, include_last_offset=True)
        ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")       # zero keys = row 0, bag 0: must not be used
        g = d_grad[:, t0 * D:t1 * D].contiguous()
        if halves:
            ws = ops.embbag_backward_prepare(workspace=ws, **kw)
            ops.embbag_backward_apply(grad=g, workspace=ws, lr=0.05, **kw)
        else:
            ops.embbag_backward_update(grad=g, workspace=ws, lr=0.05, **kw)
    scale = float(np.max(np.abs(w_ref)))
    assert np.max(np.abs(dw.cpu().numpy() - w_ref)) <= 1e-5 * scale
