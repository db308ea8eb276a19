"""Host-side mirror of the reference's `class cache` (include/data.h:242-298)
over the HBM page cache of libdlrm_amd.so.

    cache(table_num, size, dimention)      -> EmbeddingCache(table_rows, dim, ...)
    write()                                -> write()            (uniform [0,1) tables in the store)
    get_emb_vectors(ids, table_ID)         -> get_emb_vectors(ids, table_offsets)  (all tables of a batch)
    Reclaim_vector_page / reclaim_emb_cache-> reclaim()
    page_count, cache_size, write_buf[i]   -> page_count, cache_size, counters()
    read_page(i) / write_page(i)           -> counters(reset=True)

The backing store is pinned host DRAM (one [rows, dim] array per table, exposed
as numpy arrays); resident pages live in a torch CUDA tensor (`pool`) that the
EmbeddingBag operators use directly as their weight matrix.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check


class CacheConfig(C.Structure):
    _fields_ = [
        ("num_tables", C.c_int32), ("dim", C.c_int32), ("row_dtype", C.c_int32), ("rows_per_page", C.c_int32),
        ("evict_group", C.c_int32),
        ("table_rows", C.POINTER(C.c_int64)),
        ("h_table_store", C.POINTER(C.c_void_p)),
        ("num_slots", C.c_int64), ("cache_size", C.c_int64),
        ("d_slot_pool", C.c_void_p),
        ("max_lookup_ids", C.c_int64),
        ("writeback_dirty_only", C.c_int32),
        ("h_row_state_store", C.POINTER(C.c_void_p)),
        ("d_row_state_pool", C.c_void_p),
    ]


class CacheState(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("page_count", "cache_size", "ring_head", "ring_tail", "epoch", "error_flags",
                                         "last_lookup_pages_read", "total_pages", "num_slots")]


COUNTER_NAMES = ("read", "write", "hit", "miss", "evicted", "resident", "written_back")


class EmbeddingCache:
    def __init__(self, table_rows, dim, cache_size, num_slots, rows_per_page=None, evict_group=8,
                 row_dtype=torch.float32, device="cuda", dirty_only=False, max_lookup_ids=0, row_state=False):
        """row_state: every row carries one float of optimizer state (row-wise Adagrad) that is read, kept and
        written back with its page (`state_pool` on the device, `state_store[t]` on the host).
        dirty_only: evicted pages are copied back to the store only when mark_dirty() has flagged them
        (the reference writes every victim back).  max_lookup_ids: most ids one get_emb_vectors call carries;
        every buffer is then allocated here and no later call allocates."""
        self.L = _capi.lib()
        self.L.dlrm_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.dlrm_host_free.argtypes = [C.c_void_p]
        self.table_rows = [int(r) for r in table_rows]
        self.T, self.dim = len(self.table_rows), dim
        esz = 4 if row_dtype == torch.float32 else 2
        self.vpo = rows_per_page if rows_per_page else 4096 // (dim * esz)   # include/data.h:211
        self.evict_group = evict_group
        self.num_slots = num_slots
        self.device = torch.device(device)
        self._ptrs, self._state_ptrs = [], []
        self.store, self.state_store = [], []
        npdt = np.float32 if row_dtype == torch.float32 else np.uint16
        for r in self.table_rows:
            p = C.c_void_p()
            check(self.L.dlrm_host_alloc(C.byref(p), C.c_size_t(r * dim * esz)))
            self._ptrs.append(p)
            buf = (C.c_char * (r * dim * esz)).from_address(p.value)
            self.store.append(np.frombuffer(buf, dtype=npdt).reshape(r, dim))
            if row_state:
                q = C.c_void_p()
                check(self.L.dlrm_host_alloc(C.byref(q), C.c_size_t(r * 4)))
                C.memset(q, 0, r * 4)
                self._state_ptrs.append(q)
                self.state_store.append(np.frombuffer((C.c_char * (r * 4)).from_address(q.value), dtype=np.float32))
        self.state_pool = (torch.zeros(num_slots * self.vpo, dtype=torch.float32, device=self.device)
                           if row_state else None)
        self.pool = torch.zeros((num_slots * self.vpo, dim), dtype=row_dtype, device=self.device)
        cfg = CacheConfig()
        cfg.num_tables, cfg.dim = self.T, dim
        cfg.row_dtype = _capi.F32 if row_dtype == torch.float32 else _capi.BF16
        cfg.rows_per_page, cfg.evict_group = self.vpo, evict_group
        self._rows_arr = (C.c_int64 * self.T)(*self.table_rows)
        self._store_arr = (C.c_void_p * self.T)(*[p.value for p in self._ptrs])
        cfg.table_rows = self._rows_arr
        cfg.h_table_store = self._store_arr
        cfg.num_slots, cfg.cache_size = num_slots, cache_size
        cfg.d_slot_pool = self.pool.data_ptr()
        cfg.max_lookup_ids, cfg.writeback_dirty_only = int(max_lookup_ids), int(bool(dirty_only))
        if row_state:
            self._state_arr = (C.c_void_p * self.T)(*[q.value for q in self._state_ptrs])
            cfg.h_row_state_store = self._state_arr
            cfg.d_row_state_pool = self.state_pool.data_ptr()
        self.h = C.c_void_p()
        check(self.L.dlrm_cache_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize()
            self.reclaim_finish()
            torch.cuda.synchronize()
            self.L.dlrm_cache_destroy(self.h)
            self.h = None
            self.store, self.state_store = [], []
            for p in self._ptrs + self._state_ptrs:
                self.L.dlrm_host_free(p)
            self._ptrs, self._state_ptrs = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def write(self, seed=0, table_ids=None):
        """embedding_table_init::write(): every table uniform [0,1).  table_ids: the global id of
        each table (its random stream is keyed by it); default 0..T-1."""
        for t, p in enumerate(self._ptrs):
            n = self.table_rows[t] * self.dim
            dt = _capi.F32 if self.pool.dtype == torch.float32 else _capi.BF16
            gid = t if table_ids is None else table_ids[t]
            check(self.L.dlrm_table_init_uniform(p, dt, C.c_int64(n), C.c_uint64((seed << 20) + 7919 * (gid + 1)),
                                                 C.c_int64(0), self._stream()))
        torch.cuda.synchronize()

    def get_emb_vectors(self, ids, table_offsets):
        """ids: device int32/int64 tensor, ids of table t at [table_offsets[t], table_offsets[t+1]).
        Returns the row of every id inside `pool`."""
        assert ids.is_cuda and ids.is_contiguous()
        offs = (C.c_int64 * (self.T + 1))(*[int(x) for x in table_offsets])
        rows = torch.empty_like(ids)
        dt = _capi.I32 if ids.dtype == torch.int32 else _capi.I64
        check(self.L.dlrm_cache_lookup(self.h, C.c_void_p(ids.data_ptr()), dt, offs, C.c_void_p(rows.data_ptr()),
                                       self._stream()))
        return rows

    def reclaim(self):
        self.reclaim_finish()           # a pending first half belongs to an earlier step
        check(self.L.dlrm_cache_reclaim(self.h, self._stream()))

    def reclaim_table(self, table):
        """cache::Reclaim_vector_page(table_ID): the evict_group least recently used pages of one table leave."""
        self.reclaim_finish()
        check(self.L.dlrm_cache_reclaim_table(self.h, int(table), self._stream()))

    def set_protect_age(self, age):
        """Pages used in the last `age` lookup calls are never evicted by the reclaims that follow."""
        check(self.L.dlrm_cache_set_protect_age(self.h, int(age)))

    def set_async_writeback(self, max_pages):
        """A reclaim with at most max_pages victims parks them in HBM and lets a stream of the cache's own carry them
        to the store beside the next lookup's misses (0: write back inside the reclaim, the default)."""
        check(self.L.dlrm_cache_set_async_writeback(self.h, C.c_int64(int(max_pages)), self._stream()))

    def mark_dirty(self, rows):
        """rows: what get_emb_vectors returned for ids whose rows are about to be updated."""
        dt = _capi.I32 if rows.dtype == torch.int32 else _capi.I64
        check(self.L.dlrm_cache_mark_dirty(self.h, C.c_void_p(rows.data_ptr()), dt, C.c_int64(rows.numel()),
                                           self._stream()))

    def reclaim_begin(self):
        """First half of reclaim (choice of victims + asynchronous read-back of their count)."""
        check(self.L.dlrm_cache_reclaim_begin(self.h, self._stream()))
        self._pending = True

    def reclaim_finish(self):
        """Second half (write-back + eviction); waits for the first half's read-back only.  No lookup
        may be queued between the two halves."""
        if getattr(self, "_pending", False):
            check(self.L.dlrm_cache_reclaim_finish(self.h, self._stream()))
            self._pending = False

    def flush(self):
        self.reclaim_finish()
        check(self.L.dlrm_cache_flush(self.h, self._stream()))
        torch.cuda.synchronize()

    def state(self):
        s = CacheState()
        check(self.L.dlrm_cache_state(self.h, C.byref(s), self._stream()))
        return s

    @property
    def page_count(self):
        return self.state().page_count

    @property
    def cache_size(self):
        return self.state().cache_size

    @cache_size.setter
    def cache_size(self, v):
        check(self.L.dlrm_cache_set_cache_size(self.h, C.c_int64(int(v)), self._stream()))

    def counters(self, reset=False):
        """dict name -> int64[T]; reset zeroes the read/write page counters
        (cache::read_page / write_page)."""
        out = np.zeros((self.T, len(COUNTER_NAMES)), dtype=np.int64)
        check(self.L.dlrm_cache_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), int(reset), self._stream()))
        return {n: out[:, i].copy() for i, n in enumerate(COUNTER_NAMES)}

    def page_map(self, table):
        n = (self.table_rows[table] + self.vpo - 1) // self.vpo
        out = np.empty(n, dtype=np.int32)
        check(self.L.dlrm_cache_export_page_map(self.h, table, out.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()))
        return out
