"""Criteo binary reader: the device decode (dlrm_criteo_decode) against its numpy restatement on a
synthetic file in the MLPerf-DLRM record layout (no dataset ships with the reference; the record layout
is the public one: int32 label, 13 int32 counters, 26 int32 hashed ids).  Integer outputs bit-exact,
log-transformed dense features to 1e-6 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def decode_reference(records, n_dense, n_sparse, table_rows=None):
    """numpy restatement of dlrm_criteo_decode: label, log(1 + max(x, 0)) dense block, ids as unsigned modulo the
    table's rows, table-major with one id per bag."""
    rec = np.asarray(records, dtype=np.int32).reshape(-1, 1 + n_dense + n_sparse)
    n = rec.shape[0]
    labels = rec[:, 0].astype(np.float32)
    dense = np.log1p(np.maximum(rec[:, 1:1 + n_dense], 0).astype(np.float32)).astype(np.float32)
    ids = rec[:, 1 + n_dense:].astype(np.uint32).astype(np.int64)
    if table_rows is not None:
        ids = ids % np.asarray(table_rows, dtype=np.int64)[None, :]
    return dense, labels, ids.T.reshape(-1).copy(), np.arange(n_sparse * n + 1, dtype=np.int64)


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
