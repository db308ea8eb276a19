"""numpy restatement of the input path: decoding of Criteo records in the MLPerf-DLRM binary layout (single-hot
and fixed-hotness multi-hot) and the regrouping of ids after the index exchange of a sharded run.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  PARITY UNPINNED: the reference has no real-data reader --
its MyDataset (dlrm_libtorch/src/dlrmtest.cpp:299-334) draws float-encoded synthetic ids, which create_emb
(:53-63) casts back to int and splits by table -- and no multi-process run, so there is no reference output,
fixture or test for either function.  What is restated is the public record layout (int32 label, 13 int32
counters, 26 int32 hashed ids; DLRM-v2 multi-hot: a fixed number of ids per table) and the transformations of
MLPerf DLRM's data loader: log(1 + max(x, 0)) on the counters, ids taken as unsigned modulo the table's rows."""
import numpy as np


def decode(records, n_dense, n_sparse, table_rows=None, table_order=None, hotness=None):
    """records: int32 [n, 1 + n_dense + sum(hotness)].  Returns (dense [n, n_dense] float32, labels [n] float32,
    indices int64 table-major in `table_order` over the n samples (bag (t', b) = hotness[order[t']] ids),
    offsets int64 [n_sparse*n + 1])."""
    hot = [1] * n_sparse if hotness is None else [int(h) for h in hotness]
    first = np.concatenate([[0], np.cumsum(hot)]).astype(np.int64)
    rec = np.asarray(records, dtype=np.int32).reshape(-1, 1 + n_dense + int(first[-1]))
    n = rec.shape[0]
    labels = rec[:, 0].astype(np.float32)
    dense = np.log1p(np.maximum(rec[:, 1:1 + n_dense], 0).astype(np.float32)).astype(np.float32)
    order = list(range(n_sparse)) if table_order is None else [int(t) for t in table_order]
    idx, off, pos = [], [], 0
    for t in order:
        ids = rec[:, 1 + n_dense + first[t]:1 + n_dense + first[t + 1]].astype(np.uint32).astype(np.int64)
        if table_rows is not None:
            ids = ids % int(table_rows[t])
        idx.append(ids.reshape(-1))
        off.append(pos + np.arange(n, dtype=np.int64) * hot[t])
        pos += n * hot[t]
    off.append(np.asarray([pos], dtype=np.int64))
    return dense, labels, np.concatenate(idx), np.concatenate(off)


def regroup(recv, world, local_batch, hotness, offset_base=0):
    """recv: ids from every rank r, [r][table][sample][hot] flattened; hotness: ids per bag of each of this rank's
    tables.  Returns (indices table-major over the global batch, offsets)."""
    hot = [int(h) for h in hotness]
    per_sample = sum(hot)
    B, Bg = local_batch, local_batch * world
    recv = np.asarray(recv).reshape(world, per_sample * B)
    idx, off, first = [], [], 0
    for h in hot:
        idx.append(np.concatenate([recv[r, first * B:(first + h) * B] for r in range(world)]))
        off.append(offset_base + first * Bg + np.arange(Bg, dtype=np.int64) * h)
        first += h
    off.append(np.asarray([offset_base + per_sample * Bg], dtype=np.int64))
    if not idx:
        return np.zeros(0, recv.dtype), np.concatenate(off)
    return np.concatenate(idx), np.concatenate(off)
