"""Reader for Criteo click logs in the MLPerf-DLRM binary layout (`*_processed_*.bin` / `day_N.bin`
after preprocessing: per sample 1 + 13 + 26 little-endian int32; multi-hot variant: hotness[t] int32 ids per
table instead of one).  The file is memory-mapped; a batch's raw records go through a pinned staging buffer to
the device and are decoded there (dlrm_criteo_decode), so the host does no per-element work.  Stands where the
reference has MyDataset (src/dlrmtest.cpp:299-334).

    rd = CriteoBinReader(path, batch=65536, table_rows=cfg.table_rows)
    dense, labels, indices, offsets = rd.batch(i)        # device tensors, valid until batch(i + 2)

Sharded runs: every rank reads its slice of the global batch (`rank`, `world`) with
`table_order=plan.feature_order`; `DLRM.exchange_ids(indices)` then hands every rank the ids of its tables for
the global batch.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check


class CriteoBinReader:
    def __init__(self, path, batch, table_rows=None, n_dense=13, n_sparse=26, idx_dtype=torch.int32, device="cuda",
                 dense_ld=None, hotness=None, table_order=None, rank=0, world=1):
        """batch: samples per rank and step.  hotness: ids per sample of every table (default 1 each).
        table_order: output table t' is record table table_order[t'].  rank / world: step i covers records
        [i*world*batch, (i+1)*world*batch) and this reader returns the slice of `rank`."""
        self.L = _capi.lib()
        self.n_dense, self.n_sparse, self.batch_size = n_dense, n_sparse, batch
        self.rank, self.world = rank, world
        self.hotness = [1] * n_sparse if hotness is None else [int(h) for h in hotness]
        assert len(self.hotness) == n_sparse and min(self.hotness) >= 1
        self.ids_per_sample = sum(self.hotness)
        self.fields = 1 + n_dense + self.ids_per_sample
        self.mm = np.memmap(path, dtype="<i4", mode="r")
        if self.mm.size % self.fields:
            raise ValueError("%s is not a whole number of %d-field records" % (path, self.fields))
        self.n_samples = self.mm.size // self.fields
        self.dev = torch.device(device)
        self.idx_dtype = idx_dtype
        self.dense_ld = dense_ld or (n_dense + 3) // 4 * 4
        self.rows = None
        if table_rows is not None:
            self.rows = torch.tensor([int(r) for r in table_rows], dtype=torch.int64, device=self.dev)
        self.hot_off = None
        if hotness is not None:
            ho = [0]
            for h in self.hotness:
                ho.append(ho[-1] + h)
            self.hot_off = torch.tensor(ho, dtype=torch.int32, device=self.dev)
        self.order = None
        if table_order is not None:
            assert sorted(int(t) for t in table_order) == list(range(n_sparse))
            self.order = torch.tensor([int(t) for t in table_order], dtype=torch.int32, device=self.dev)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        # two sets of buffers: batch i+1 can be staged while batch i is in use
        self.slots = []
        for _ in range(2):
            self.slots.append(dict(
                host=torch.empty(batch * self.fields, dtype=torch.int32).pin_memory(),
                raw=torch.empty(batch * self.fields, dtype=torch.int32, device=self.dev),
                dense=torch.zeros((batch, self.dense_ld), dtype=torch.float32, device=self.dev),
                labels=torch.empty(batch, dtype=torch.float32, device=self.dev),
                idx=torch.empty(self.ids_per_sample * batch, dtype=idx_dtype, device=self.dev),
                off=torch.empty(n_sparse * batch + 1, dtype=idx_dtype, device=self.dev),
                done=torch.cuda.Event()))

    def __len__(self):
        return self.n_samples // (self.batch_size * self.world)     # whole global batches (the tail is dropped)

    def batch_records(self, i):
        s = ((i % len(self)) * self.world + self.rank) * self.batch_size * self.fields
        return self.mm[s:s + self.batch_size * self.fields]

    def batch(self, i):
        """Stages and decodes batch i on the reader's copy stream; the caller's current stream waits for it."""
        sl = self.slots[i % 2]
        B = self.batch_size
        self.copy_stream.wait_stream(torch.cuda.current_stream())     # earlier users of this slot are done
        sl["host"].numpy()[:] = self.batch_records(i)
        with torch.cuda.stream(self.copy_stream):
            sl["raw"].copy_(sl["host"], non_blocking=True)
            check(self.L.dlrm_criteo_decode(
                C.c_void_p(sl["raw"].data_ptr()), C.c_int64(B), self.n_dense, self.n_sparse,
                C.c_void_p(0 if self.rows is None else self.rows.data_ptr()),
                C.c_void_p(0 if self.order is None else self.order.data_ptr()),
                C.c_void_p(0 if self.hot_off is None else self.hot_off.data_ptr()), C.c_int32(self.ids_per_sample),
                C.c_void_p(sl["dense"].data_ptr()),
                C.c_int64(self.dense_ld), C.c_void_p(sl["labels"].data_ptr()), C.c_void_p(sl["idx"].data_ptr()),
                C.c_void_p(sl["off"].data_ptr()), _capi.I32 if self.idx_dtype == torch.int32 else _capi.I64,
                C.c_void_p(self.copy_stream.cuda_stream)))
            sl["done"].record()
        torch.cuda.current_stream().wait_event(sl["done"])
        return sl["dense"][:, :self.n_dense], sl["labels"], sl["idx"], sl["off"]
