"""ctypes binding of the CPU oracle (oracle/dlrm_oracle.c, oracle/vec_cache_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under dlrm_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

POOL_SUM, POOL_MEAN = 0, 1
DUP_ACCUMULATE, DUP_LAST_WINS = 0, 1
OPT_SGD, OPT_ADAGRAD_ROWWISE = 0, 1


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("dlrm_oracle.c", "vec_cache_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_pagecache_create.restype = C.c_void_p
        _lib.orc_pagecache_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64]
        for name in ("orc_pagecache_get", "orc_pagecache_reclaim", "orc_pagecache_page_count",
                     "orc_pagecache_cache_size", "orc_pagecache_lru_len", "orc_pagecache_lru_order",
                     "orc_pagecache_slot_of", "orc_pagecache_ring_head", "orc_pagecache_ring_tail"):
            getattr(_lib, name).restype = C.c_int64
        _lib.orc_mse_loss.restype = C.c_float
        _lib.orc_bce_logits_loss.restype = C.c_float
        _lib.orc_veccache_create.restype = C.c_void_p
        _lib.orc_veccache_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64]
        for name in ("orc_veccache_get", "orc_veccache_reclaim", "orc_veccache_page_count", "orc_veccache_list_len",
                     "orc_veccache_pending", "orc_veccache_written_pages", "orc_veccache_fifo_order",
                     "orc_veccache_pending_rows"):
            getattr(_lib, name).restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def table_init_uniform(n, seed, first=0):
    w = np.empty(n, dtype=np.float32)
    lib().orc_table_init_uniform(_p(w), C.c_int64(n), C.c_uint64(seed), C.c_int64(first))
    return w


def f32_to_bf16(x):
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_f32_to_bf16(_p(x), _p(out), C.c_int64(x.size))
    return out


def bf16_to_f32(x):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.empty(x.shape, dtype=np.float32)
    lib().orc_bf16_to_f32(_p(x), _p(out), C.c_int64(x.size))
    return out


def embbag_forward(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None,
                   pooling=POOL_SUM, include_last_offset=False, out=None, out_ld=None):
    """weight: [rows, D] float32 or uint16 (bf16 bits). Returns [batch, T*D] float32."""
    row_bf16 = int(weight.dtype == np.uint16)
    weight = np.ascontiguousarray(weight)
    dim = weight.shape[1]
    indices, offsets = _i64(indices), _i64(offsets)
    tro = None if table_row_offsets is None else _i64(table_row_offsets)
    if out is None:
        out_ld = num_tables * dim
        out = np.zeros((batch, out_ld), dtype=np.float32)
    rc = lib().orc_embbag_forward(_p(weight), row_bf16, _p(tro), num_tables, dim, _p(indices),
                                  _p(offsets), C.c_int64(batch), C.c_int64(indices.size),
                                  int(include_last_offset), pooling, _p(out), C.c_int64(out_ld))
    if rc:
        raise ValueError("orc_embbag_forward rc=%d" % rc)
    return out


def embbag_backward_update(weight, indices, offsets, batch, grad, num_tables=1,
                           table_row_offsets=None, pooling=POOL_SUM, include_last_offset=False,
                           optimizer=OPT_SGD, dup_mode=DUP_ACCUMULATE, lr=0.01, eps=1e-8,
                           adagrad_state=None, grad_ld=None):
    """Updates weight (and adagrad_state) in place."""
    assert weight.flags.c_contiguous and grad.dtype == np.float32
    row_bf16 = int(weight.dtype == np.uint16)
    dim = weight.shape[1]
    indices, offsets = _i64(indices), _i64(offsets)
    tro = None if table_row_offsets is None else _i64(table_row_offsets)
    if grad_ld is None:
        grad_ld = grad.shape[1]
    rc = lib().orc_embbag_backward_update(
        _p(weight), row_bf16, _p(tro), num_tables, dim, _p(indices), _p(offsets), C.c_int64(batch),
        C.c_int64(indices.size), int(include_last_offset), pooling, _p(grad), C.c_int64(grad_ld),
        optimizer, dup_mode, C.c_float(lr), C.c_float(eps), _p(adagrad_state))
    if rc:
        raise ValueError("orc_embbag_backward_update rc=%d" % rc)


def linear_forward(x, w, bias, relu):
    x, w = _f32(x), _f32(w)
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), dtype=np.float32)
    b = None if bias is None else _f32(bias)
    lib().orc_linear_forward(_p(x), C.c_int64(K), _p(w), _p(b), _p(y), C.c_int64(N), C.c_int64(M), N, K, int(relu))
    return y


def linear_backward(x, w, y, dy, relu, need_dx=True):
    x, w, y, dy = _f32(x), _f32(w), _f32(y), _f32(dy)
    M, K = x.shape
    N = w.shape[0]
    dx = np.empty((M, K), dtype=np.float32) if need_dx else None
    dw = np.empty((N, K), dtype=np.float32)
    db = np.empty((N,), dtype=np.float32)
    lib().orc_linear_backward(_p(x), C.c_int64(K), _p(w), _p(y), C.c_int64(N), _p(dy), C.c_int64(N),
                              _p(dx), C.c_int64(K), _p(dw), _p(db), C.c_int64(M), N, K, int(relu))
    return dx, dw, db


def sgd_step(w, g, lr):
    assert w.dtype == np.float32 and w.flags.c_contiguous
    g = _f32(g)
    lib().orc_sgd_step(_p(w), _p(g), C.c_int64(w.size), C.c_float(lr))


def interact_dot_forward(x):
    x = _f32(x)
    B, F, D = x.shape
    n_out = D + F * (F - 1) // 2
    out = np.empty((B, n_out), dtype=np.float32)
    lib().orc_interact_dot_forward(_p(x), C.c_int64(B), F, D, _p(out), C.c_int64(n_out))
    return out


def interact_dot_backward(x, dout):
    x, dout = _f32(x), _f32(dout)
    B, F, D = x.shape
    dx = np.empty_like(x)
    lib().orc_interact_dot_backward(_p(x), _p(dout), C.c_int64(dout.shape[1]), C.c_int64(B), F, D, _p(dx))
    return dx


def mse_loss(pred, target):
    pred, target = _f32(pred).reshape(-1), _f32(target).reshape(-1)
    d = np.empty_like(pred)
    v = lib().orc_mse_loss(_p(pred), _p(target), C.c_int64(pred.size), _p(d))
    return float(v), d


def bce_logits_loss(logit, target):
    logit, target = _f32(logit).reshape(-1), _f32(target).reshape(-1)
    d = np.empty_like(logit)
    v = lib().orc_bce_logits_loss(_p(logit), _p(target), C.c_int64(logit.size), _p(d))
    return float(v), d


class PageCache:
    """Page-LRU embedding cache state machine (restates class cache,
    include/data.h:242-298, and reclaim_emb_cache, src/dlrmtest.cpp:397-412)."""

    def __init__(self, num_tables, rows_per_table, dim, cache_size, page_bytes=4096, ws_opt=8,
                 n_slots=0):
        self.L = lib()
        self.T = num_tables
        self.h = C.c_void_p(self.L.orc_pagecache_create(num_tables, C.c_int64(rows_per_table), dim,
                                                        page_bytes, ws_opt, C.c_int64(cache_size)))
        self.vpo = self.L.orc_pagecache_vpo(self.h)
        self.pages_per_table = (rows_per_table + self.vpo - 1) // self.vpo
        if n_slots:
            self.L.orc_pagecache_enable_slots(self.h, C.c_int64(n_slots))

    def __del__(self):
        try:
            self.L.orc_pagecache_destroy(self.h)
        except Exception:
            pass

    def get(self, table, ids):
        """cache::get_emb_vectors: returns the pages read, in read order."""
        ids = _i64(ids)
        miss = np.empty(max(ids.size, 1), dtype=np.int64)
        n = self.L.orc_pagecache_get(self.h, table, _p(ids), C.c_int64(ids.size), _p(miss))
        if n < 0:
            raise RuntimeError("slot pool exhausted")
        return miss[:n].copy()

    def reclaim(self):
        return self.L.orc_pagecache_reclaim(self.h)

    def set_dirty_only(self, on=True):
        self.L.orc_pagecache_set_dirty_only(self.h, int(on))

    def mark_dirty(self, table, ids):
        ids = _i64(ids)
        self.L.orc_pagecache_mark_dirty(self.h, table, _p(ids), C.c_int64(ids.size))

    def reclaim_table(self, table):
        ev = np.empty(64, dtype=np.int64)
        k = self.L.orc_pagecache_reclaim_table(self.h, table, _p(ev))
        return ev[:k].copy()

    @property
    def page_count(self):
        return self.L.orc_pagecache_page_count(self.h)

    @property
    def cache_size(self):
        return self.L.orc_pagecache_cache_size(self.h)

    @cache_size.setter
    def cache_size(self, v):
        self.L.orc_pagecache_set_cache_size(self.h, C.c_int64(v))

    def counters(self, which, reset=False):
        out = np.zeros(self.T, dtype=np.int64)
        self.L.orc_pagecache_counters(self.h, {"read": 0, "write": 1, "hit": 2, "miss": 3, "written_back": 4}[which],
                                      _p(out), int(reset))
        return out

    def lru_order(self, table):
        n = self.L.orc_pagecache_lru_len(self.h, table)
        out = np.empty(max(n, 1), dtype=np.int64)
        k = self.L.orc_pagecache_lru_order(self.h, table, _p(out), C.c_int64(n))
        return out[:k].copy()

    def slot_map(self, table):
        out = np.empty(self.pages_per_table, dtype=np.int64)
        self.L.orc_pagecache_slot_map(self.h, table, _p(out), C.c_int64(self.pages_per_table))
        return out

    def slot_of(self, table, page):
        return self.L.orc_pagecache_slot_of(self.h, table, C.c_int64(page))


class VectorCache:
    """Vector-granular FIFO cache over a log-structured page store (restates class cache2,
    include/data2.h:219-287 + src/emb_cache2.cpp, as driven by src/dlrmtest2.cpp)."""

    def __init__(self, num_tables, rows_per_table, dim, cache_size, page_bytes=4096, ws_opt=8):
        self.L = lib()
        self.T, self.rows = num_tables, rows_per_table
        self.h = C.c_void_p(self.L.orc_veccache_create(num_tables, C.c_int64(rows_per_table), dim, page_bytes,
                                                       ws_opt, C.c_int64(cache_size)))
        self.vpo = self.L.orc_veccache_vpo(self.h)

    def __del__(self):
        try:
            self.L.orc_veccache_destroy(self.h)
        except Exception:
            pass

    def get(self, table, ids):
        """cache2::get_emb_vectors: returns the pages read, in read order."""
        ids = _i64(ids)
        pages = np.empty(max(ids.size, 1), dtype=np.int64)
        n = self.L.orc_veccache_get(self.h, table, _p(ids), C.c_int64(ids.size), _p(pages))
        return pages[:n].copy()

    def reclaim(self):
        """reclaim_emb_cache of dlrmtest2: returns the number of Reclaim_vector_page calls."""
        return self.L.orc_veccache_reclaim(self.h)

    def reclaim_table(self, table):
        return self.L.orc_veccache_reclaim_table(self.h, table)

    def save(self, table, ids):
        ids = _i64(ids)
        self.L.orc_veccache_save(self.h, table, _p(ids), C.c_int64(ids.size))

    @property
    def page_count(self):
        return self.L.orc_veccache_page_count(self.h)

    def list_len(self, table):
        return self.L.orc_veccache_list_len(self.h, table)

    def pending(self, table):
        return self.L.orc_veccache_pending(self.h, table)

    def written_pages(self, table):
        return self.L.orc_veccache_written_pages(self.h, table)

    def counters(self, which, reset=False):
        out = np.zeros(self.T, dtype=np.int64)
        self.L.orc_veccache_counters(self.h, {"read": 0, "write": 1, "hit": 2, "miss": 3}[which], _p(out), int(reset))
        return out

    def export(self, table):
        """(resident, listed, cnt, loc) of every vector of the table."""
        res, lst = np.empty(self.rows, np.uint8), np.empty(self.rows, np.uint8)
        cnt, loc = np.empty(self.rows, np.int32), np.empty(self.rows, np.int32)
        self.L.orc_veccache_export(self.h, table, _p(res), _p(lst), _p(cnt), _p(loc))
        return res, lst, cnt, loc

    def fifo_order(self, table):
        n = self.list_len(table)
        out = np.empty(max(n, 1), dtype=np.int64)
        k = self.L.orc_veccache_fifo_order(self.h, table, _p(out), C.c_int64(n))
        return out[:k].copy()

    def pending_rows(self, table):
        n = self.pending(table)
        out = np.empty(max(n, 1), dtype=np.int64)
        k = self.L.orc_veccache_pending_rows(self.h, table, _p(out), C.c_int64(n))
        return out[:k].copy()
