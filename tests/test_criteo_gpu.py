"""Criteo binary reader: the device decode (dlrm_criteo_decode) and the regrouping of exchanged ids
(dlrm_ids_regroup) against their numpy restatements (oracle/criteo_oracle.py; parity unpinned: the reference
has no reader and no multi-process run) on synthetic files in the MLPerf-DLRM record layout (int32 label, 13
int32 counters, 26 int32 hashed ids; multi-hot: a fixed number of ids per table).  Integer outputs bit-exact,
log-transformed dense features to 1e-6 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def decode_reference(records, n_dense, n_sparse, table_rows=None, table_order=None, hotness=None):
    from oracle import criteo_oracle
    return criteo_oracle.decode(records, n_dense, n_sparse, table_rows, table_order, hotness)


@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("B", [1, 100, 257])
def test_reader_matches_numpy_decode(tmp_path, idx_t, B):
    from dlrm_amd.criteo import CriteoBinReader
    rng = np.random.default_rng(B)
    n = 3 * B + B // 2                                 # a tail that is dropped
    rec = np.empty((n, 40), dtype=np.int32)
    rec[:, 0] = rng.integers(0, 2, n)
    rec[:, 1:14] = rng.integers(-3, 100000, (n, 13))   # counters, a few negative (clipped to 0)
    rec[:, 14:] = rng.integers(0, 2**32, (n, 26), dtype=np.uint64).astype(np.uint32).view(np.int32)
    path = tmp_path / "day.bin"
    rec.tofile(path)
    rows = [int(x) for x in rng.integers(1, 10_000_000, 26)]
    rd = CriteoBinReader(str(path), B, table_rows=rows, idx_dtype=idx_t)
    assert len(rd) == 3
    for i in range(4):                                 # the fourth wraps around to batch 0
        dense, labels, idx, off = rd.batch(i)
        torch.cuda.synchronize()
        d_ref, l_ref, i_ref, o_ref = decode_reference(rec[(i % 3) * B:(i % 3 + 1) * B], 13, 26, rows)
        assert np.array_equal(labels.cpu().numpy(), l_ref)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), i_ref)
        assert np.array_equal(off.cpu().numpy().astype(np.int64), o_ref)
        assert np.allclose(dense.cpu().numpy(), d_ref, rtol=1e-6, atol=0)


def test_reader_feeds_a_train_step(tmp_path):
    from dlrm_amd.criteo import CriteoBinReader
    from dlrm_amd.model import DLRM, DLRMConfig
    rng = np.random.default_rng(0)
    B, T = 64, 26
    rec = rng.integers(0, 1000, (4 * B, 40)).astype(np.int32)
    rec[:, 0] = rng.integers(0, 2, 4 * B)
    path = tmp_path / "day.bin"
    rec.tofile(path)
    cfg = DLRMConfig(table_rows=[int(x) for x in rng.integers(5, 500, T)], dim=16, dense_in=13, bottom=[32, 16],
                     top=[32, 1], lr=0.1)
    rd = CriteoBinReader(str(path), B, table_rows=cfg.table_rows)
    m = DLRM(cfg, B, seed=1)
    losses = []
    for i in range(12):
        dense, labels, idx, off = rd.batch(i)
        m.load_batch(dense, labels)
        losses.append(m.train_step(idx, off).item())
    assert int(m.err_flag.item()) == 0 and all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("B", [1, 100, 300])
def test_multi_hot_records_in_plan_order(tmp_path, idx_t, B):
    """Fixed-hotness multi-hot records (tables with 1..7 ids per sample), output tables permuted (a sharding
    plan's order), two ranks each reading their slice of every global batch."""
    from dlrm_amd.criteo import CriteoBinReader
    rng = np.random.default_rng(100 + B)
    T = 9
    hot = [int(h) for h in rng.integers(1, 8, T)]
    hot[3] = 1
    order = [int(t) for t in rng.permutation(T)]
    world = 2
    n = 3 * world * B + B // 2
    fields = 1 + 13 + sum(hot)
    rec = np.empty((n, fields), dtype=np.int32)
    rec[:, 0] = rng.integers(0, 2, n)
    rec[:, 1:14] = rng.integers(-3, 100000, (n, 13))
    rec[:, 14:] = rng.integers(0, 2**32, (n, sum(hot)), dtype=np.uint64).astype(np.uint32).view(np.int32)
    path = tmp_path / "mh.bin"
    rec.tofile(path)
    rows = [int(x) for x in rng.integers(1, 10_000_000, T)]
    for rank in range(world):
        rd = CriteoBinReader(str(path), B, table_rows=rows, n_sparse=T, idx_dtype=idx_t, hotness=hot, table_order=order,
                             rank=rank, world=world)
        assert len(rd) == 3
        for i in range(4):
            dense, labels, idx, off = rd.batch(i)
            torch.cuda.synchronize()
            first = ((i % 3) * world + rank) * B
            d_ref, l_ref, i_ref, o_ref = decode_reference(rec[first:first + B], 13, T, rows, order, hot)
            assert np.array_equal(labels.cpu().numpy(), l_ref)
            assert np.array_equal(idx.cpu().numpy().astype(np.int64), i_ref)
            assert np.array_equal(off.cpu().numpy().astype(np.int64), o_ref)
            assert np.allclose(dense.cpu().numpy(), d_ref, rtol=1e-6, atol=0)


@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("world,B,hot", [(1, 37, [1, 1, 1]), (4, 50, [2, 1, 5, 3]), (8, 300, [1] * 4), (3, 1, [4])])
def test_ids_regroup_matches_restatement(idx_t, world, B, hot):
    from dlrm_amd import ops
    from oracle import criteo_oracle
    rng = np.random.default_rng(world * 1000 + B)
    recv = rng.integers(0, 2**31 - 1, world * B * sum(hot)).astype(np.int64)
    hot_off = torch.tensor(np.concatenate([[0], np.cumsum(hot)]), dtype=torch.int32, device="cuda")
    d_recv = torch.from_numpy(recv).to(idx_t).cuda()
    for base, ho in ((0, hot_off), (12345, hot_off)) + (((7, None),) if max(hot) == 1 else ()):
        idx, off = ops.ids_regroup(d_recv, world, B, len(hot), ho, base)
        torch.cuda.synchronize()
        i_ref, o_ref = criteo_oracle.regroup(recv, world, B, hot, base)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), i_ref)
        assert np.array_equal(off.cpu().numpy().astype(np.int64), o_ref)


def test_ids_regroup_refuses_offsets_beyond_32_bits():
    """32-bit ids: an offset base that would push the CSR offsets past 2^31 is refused on the host (they would wrap into
    negative offsets that the EmbeddingBag kernels then read); the same call with 64-bit ids goes through."""
    from dlrm_amd import _capi, ops
    world, B, hot = 2, 16, [3, 1]
    recv = torch.zeros(world * B * sum(hot), dtype=torch.int32, device="cuda")
    hot_off = torch.tensor([0, 3, 4], dtype=torch.int32, device="cuda")
    base = 2**31 - 100
    with pytest.raises(_capi.DlrmError, match="32-bit"):
        ops.ids_regroup(recv, world, B, len(hot), hot_off, base)
    idx, off = ops.ids_regroup(recv.to(torch.int64), world, B, len(hot), hot_off, base)
    torch.cuda.synchronize()
    assert int(off[-1].item()) == base + world * B * sum(hot)
