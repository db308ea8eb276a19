"""Host-side mirror of the reference's `class cache2` (include/data2.h:219-287) over the vector-granular
HBM cache of libdlrm_amd.so.

    cache2(table_num, size, dimention)     -> VectorCache(table_rows, dim, cache_size, num_slots, ...)
    write()                                -> write()            (uniform [0,1) tables in the store)
    get_emb_vectors(ids, table_ID)         -> get_emb_vectors(ids, table_offsets)  (all tables of a batch)
    Reclaim_vector_page / reclaim_emb_cache-> reclaim()
    save_emb's resurrection of dropped rows-> restore(ids, table_offsets)
    page_count, write_buf[i], lru_main[i]  -> page_count, counters(), list_order(t)
    read_page(i) / write_page(i)           -> counters(reset=True)

The store of a table is pinned host DRAM: the table itself ([rows, dim], exposed as numpy arrays) followed by
the region the cache writes its pages to.  Resident rows live in a torch CUDA tensor (`pool`) that the
EmbeddingBag operators use directly as their weight matrix.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check


class VCacheConfig(C.Structure):
    _fields_ = [
        ("num_tables", C.c_int32), ("dim", C.c_int32), ("rows_per_page", C.c_int32), ("evict_group", C.c_int32),
        ("table_rows", C.POINTER(C.c_int64)),
        ("h_table_store", C.POINTER(C.c_void_p)),
        ("log_pages_per_table", C.c_int64), ("num_slots", C.c_int64), ("cache_size", C.c_int64),
        ("d_slot_pool", C.c_void_p),
        ("row_dtype", C.c_int32),
        ("h_row_state_store", C.POINTER(C.c_void_p)),
        ("d_row_state_pool", C.c_void_p),
    ]


class VCacheState(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("page_count", "cache_size", "free_slots", "epoch", "error_flags",
                                         "last_lookup_pages_read", "num_slots", "total_rows")]


COUNTER_NAMES = ("read", "write", "hit", "miss", "evicted", "resident", "listed", "dropped", "pending", "written")


class VectorCache:
    def __init__(self, table_rows, dim, cache_size, num_slots, log_pages_per_table, rows_per_page=None,
                 evict_group=8, device="cuda", row_dtype=torch.float32, row_state=False):
        """row_dtype: torch.float32 (the reference's rows) or torch.bfloat16 (the store's arrays then hold the bf16
        bit patterns as uint16).  row_state: every row carries one float of optimizer state (row-wise Adagrad)
        that moves with it between store, write buffer and pool (`state_pool`, `state_store`)."""
        self.L = _capi.lib()
        self.L.dlrm_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.dlrm_host_free.argtypes = [C.c_void_p]
        self.table_rows = [int(r) for r in table_rows]
        self.T, self.dim = len(self.table_rows), dim
        assert row_dtype in (torch.float32, torch.bfloat16)
        self.row_dtype = row_dtype
        esz = 4 if row_dtype == torch.float32 else 2
        self.vpo = rows_per_page if rows_per_page else 4096 // (dim * esz)    # include/data2.h:161
        self.evict_group, self.num_slots, self.log_pages = evict_group, num_slots, log_pages_per_table
        self.device = torch.device(device)
        self._ptrs, self.store = [], []
        self._state_ptrs, self.state_store = [], []
        page_bytes = self.vpo * dim * esz
        for r in self.table_rows:
            n_init = (r + self.vpo - 1) // self.vpo
            nbytes = (n_init + log_pages_per_table) * page_bytes
            p = C.c_void_p()
            check(self.L.dlrm_host_alloc(C.byref(p), C.c_size_t(nbytes)))
            self._ptrs.append(p)
            buf = (C.c_char * (r * dim * esz)).from_address(p.value)
            self.store.append(np.frombuffer(buf, dtype=np.float32 if esz == 4 else np.uint16).reshape(r, dim))
            if row_state:
                n_state = (n_init + log_pages_per_table) * self.vpo
                q = C.c_void_p()
                check(self.L.dlrm_host_alloc(C.byref(q), C.c_size_t(n_state * 4)))
                C.memset(q, 0, n_state * 4)
                self._state_ptrs.append(q)
                self.state_store.append(np.frombuffer((C.c_char * (n_state * 4)).from_address(q.value), dtype=np.float32))
        self.pool = torch.zeros((num_slots, dim), dtype=row_dtype, device=self.device)
        self.state_pool = torch.zeros(num_slots, dtype=torch.float32, device=self.device) if row_state else None
        cfg = VCacheConfig()
        cfg.num_tables, cfg.dim, cfg.rows_per_page, cfg.evict_group = self.T, dim, self.vpo, evict_group
        self._rows_arr = (C.c_int64 * self.T)(*self.table_rows)
        self._store_arr = (C.c_void_p * self.T)(*[p.value for p in self._ptrs])
        cfg.table_rows, cfg.h_table_store = self._rows_arr, self._store_arr
        cfg.log_pages_per_table, cfg.num_slots, cfg.cache_size = log_pages_per_table, num_slots, cache_size
        cfg.d_slot_pool = self.pool.data_ptr()
        cfg.row_dtype = _capi.F32 if esz == 4 else _capi.BF16
        if row_state:
            self._state_arr = (C.c_void_p * self.T)(*[q.value for q in self._state_ptrs])
            cfg.h_row_state_store = self._state_arr
            cfg.d_row_state_pool = self.state_pool.data_ptr()
        self.h = C.c_void_p()
        check(self.L.dlrm_vcache_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize()
            self.L.dlrm_vcache_destroy(self.h)
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

    def _offs(self, table_offsets):
        return (C.c_int64 * (self.T + 1))(*[int(x) for x in table_offsets])

    @staticmethod
    def _dt(ids):
        assert ids.is_cuda and ids.is_contiguous() and ids.dtype in (torch.int32, torch.int64)
        return _capi.I32 if ids.dtype == torch.int32 else _capi.I64

    def write(self, seed=0):
        """embedding_table_init2::write(): every table uniform [0,1)."""
        self.write_tables(seed, range(self.T))

    def write_tables(self, seed, table_ids):
        """table_ids: the global id of each table (its random stream is keyed by it)."""
        for t, p in enumerate(self._ptrs):
            n = self.table_rows[t] * self.dim
            check(self.L.dlrm_table_init_uniform(p, _capi.F32 if self.row_dtype == torch.float32 else _capi.BF16,
                                                 C.c_int64(n),
                                                 C.c_uint64((seed << 20) + 7919 * (table_ids[t] + 1)),
                                                 C.c_int64(0), self._stream()))
        torch.cuda.synchronize()

    def get_emb_vectors(self, ids, table_offsets):
        """ids of table t at [table_offsets[t], table_offsets[t+1]), in request order.  Returns the row of
        every id inside `pool`."""
        rows = torch.empty_like(ids)
        check(self.L.dlrm_vcache_lookup(self.h, C.c_void_p(ids.data_ptr()), self._dt(ids), self._offs(table_offsets),
                                        C.c_void_p(rows.data_ptr()), self._stream()))
        return rows

    def reclaim(self):
        check(self.L.dlrm_vcache_reclaim(self.h, self._stream()))

    def restore(self, ids, table_offsets, want_rows=False):
        rows = torch.empty_like(ids) if want_rows else None
        check(self.L.dlrm_vcache_restore(self.h, C.c_void_p(ids.data_ptr()), self._dt(ids), self._offs(table_offsets),
                                         C.c_void_p(rows.data_ptr() if want_rows else 0), self._stream()))
        return rows

    def read_rows(self, ids, table_offsets, with_state=False):
        """Current value of every id's row wherever it lives (and, with_state, its optimizer state)."""
        out = torch.empty((ids.numel(), self.dim), dtype=self.row_dtype, device=ids.device)
        st = torch.zeros(ids.numel(), dtype=torch.float32, device=ids.device) if with_state else None
        check(self.L.dlrm_vcache_read_rows(self.h, C.c_void_p(ids.data_ptr()), self._dt(ids), self._offs(table_offsets),
                                           C.c_void_p(out.data_ptr()), C.c_void_p(st.data_ptr() if with_state else 0),
                                           self._stream()))
        return (out, st) if with_state else out

    def state(self):
        s = VCacheState()
        check(self.L.dlrm_vcache_state(self.h, C.byref(s), self._stream()))
        return s

    @property
    def page_count(self):
        return self.state().page_count

    def set_cache_size(self, v):
        check(self.L.dlrm_vcache_set_cache_size(self.h, C.c_int64(int(v)), self._stream()))

    def counters(self, reset=False):
        """dict name -> int64[T]; reset zeroes the read/write page counters."""
        out = np.zeros((self.T, len(COUNTER_NAMES)), dtype=np.int64)
        check(self.L.dlrm_vcache_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), int(reset), self._stream()))
        return {n: out[:, i].copy() for i, n in enumerate(COUNTER_NAMES)}

    def export_table(self, table):
        """(slot, page, pending position) of every row of the table."""
        n = self.table_rows[table]
        slot, page, pend = (np.empty(n, dtype=np.int32) for _ in range(3))
        p32 = C.POINTER(C.c_int32)
        check(self.L.dlrm_vcache_export_table(self.h, table, slot.ctypes.data_as(p32), page.ctypes.data_as(p32),
                                              pend.ctypes.data_as(p32), self._stream()))
        return slot, page, pend

    def export_slots(self):
        """(global row, list key, hit count) of every slot."""
        row = np.empty(self.num_slots, dtype=np.int32)
        key = np.empty(self.num_slots, dtype=np.uint64)
        cnt = np.empty(self.num_slots, dtype=np.int32)
        check(self.L.dlrm_vcache_export_slots(self.h, row.ctypes.data_as(C.POINTER(C.c_int32)),
                                              key.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              cnt.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()))
        return row, key, cnt

    def table_view(self, table):
        """Per row of the table: resident (bool), listed (bool), hit count, page; and the listed rows oldest
        first -- the state the reference keeps in emb_vec / lru_main / hot_table."""
        slot, page, pend = self.export_table(table)
        srow, skey, scnt = self.export_slots()
        resident = slot >= 0
        s = np.where(resident, slot, 0)
        listed = resident & (skey[s] != 0)
        cnt = np.where(listed, scnt[s], 0)
        rows = np.nonzero(listed)[0]
        order = rows[np.argsort(skey[slot[rows]], kind="stable")]
        return resident, listed, cnt, page, order, pend
