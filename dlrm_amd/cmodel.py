"""ctypes mirror of the C ABI's whole-model handle (dlrm_model_*, include/dlrm_amd.h): the train / inference step
as a C or C++ host drives it -- placement, streams, exchanges and the order of the operators all live in
libdlrm_amd.so (csrc/model.hip); this module only converts arguments.  Used by the tests and by
`bench.py --host c`; dlrm_amd.model is the Python-hosted twin of the same schedule."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check

INTERACT = {"dot": 0, "cat": 1}
LOSS = {"bce": 0, "mse": 1}


class ModelConfig(C.Structure):
    _fields_ = [
        ("num_tables", C.c_int32), ("table_rows", C.POINTER(C.c_int64)), ("dim", C.c_int32), ("dense_in", C.c_int32),
        ("n_bottom", C.c_int32), ("bottom", C.POINTER(C.c_int32)), ("n_top", C.c_int32), ("top", C.POINTER(C.c_int32)),
        ("interaction", C.c_int32), ("loss", C.c_int32), ("final_relu", C.c_int32), ("pooling", C.c_int32),
        ("row_dtype", C.c_int32), ("idx_dtype", C.c_int32), ("emb_optimizer", C.c_int32), ("gemm_mode", C.c_int32),
        ("lr", C.c_float), ("eps", C.c_float),
        ("local_batch", C.c_int64), ("rank", C.c_int32), ("world", C.c_int32), ("comm", C.c_void_p),
        ("table_owner", C.POINTER(C.c_int32)), ("force_exchange", C.c_int32),
        ("overlap", C.c_int32), ("presort", C.c_int32), ("wgrad_stream", C.c_int32),
        ("seed", C.c_uint64), ("init_tables", C.c_int32), ("one_id_per_bag", C.c_int32), ("graph_steps", C.c_int32),
        ("comm_dense", C.c_void_p),
    ]


class ModelInfo(C.Structure):
    _fields_ = [
        ("d_params", C.c_void_p), ("d_grads", C.c_void_p), ("n_params", C.c_int64),
        ("d_weight", C.c_void_p), ("total_rows", C.c_int64), ("d_opt_state", C.c_void_p),
        ("num_local_tables", C.c_int32), ("num_whole_tables", C.c_int32),
        ("d_pred", C.c_void_p), ("d_loss", C.c_void_p), ("d_error_flag", C.c_void_p),
        ("d_dense", C.c_void_p), ("dense_ld", C.c_int64), ("d_labels", C.c_void_p),
        ("n_bottom", C.c_int32), ("n_top", C.c_int32),
    ]


def _view(ptr, shape, dtype, device):
    """torch view of library-owned device memory (no copy, no ownership)."""
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=dtype, device=device)
    esz = torch.empty((), dtype=dtype).element_size()
    iface = {"shape": (n * esz,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    holder = type("_Mem", (), {"__cuda_array_interface__": iface})()
    return torch.as_tensor(holder, device=device).view(dtype).view(shape)


class CModel:
    """cfg: dlrm_amd.model.DLRMConfig.  comm: dlrm_amd.comm.Communicator (sharded runs); comm_dense: a second one over
    the same ranks for the all-reduce of the dense gradients (it then runs beside the backward exchange)."""

    def __init__(self, cfg, batch, rank=0, world=1, comm=None, table_owner=None, seed=0, force_exchange=False,
                 overlap=-1, presort=1, wgrad_stream=-1, idx_dtype=torch.int32, device="cuda", gemm_mode=0,
                 one_id_per_bag=False, graph_steps=False, comm_dense=None):
        self.L = _capi.lib()
        self.cfg, self.B, self.rank, self.world = cfg, batch, rank, world
        self.dev = torch.device(device)
        c = ModelConfig()
        T = cfg.num_tables
        self._rows = (C.c_int64 * T)(*cfg.table_rows)
        self._bot = (C.c_int32 * len(cfg.bottom))(*cfg.bottom)
        self._top = (C.c_int32 * len(cfg.top))(*cfg.top)
        c.num_tables, c.table_rows, c.dim, c.dense_in = T, self._rows, cfg.dim, cfg.dense_in
        c.n_bottom, c.bottom, c.n_top, c.top = len(cfg.bottom), self._bot, len(cfg.top), self._top
        c.interaction, c.loss = INTERACT[cfg.interaction], LOSS[cfg.loss]
        c.final_relu = int(cfg.final_relu)
        c.pooling = _capi.POOL_MEAN if cfg.pooling == "mean" else _capi.POOL_SUM
        c.row_dtype = _capi.BF16 if cfg.row_dtype == "bf16" else _capi.F32
        c.idx_dtype = _capi.I32 if idx_dtype == torch.int32 else _capi.I64
        c.emb_optimizer = _capi.OPT_ADAGRAD_ROWWISE if cfg.emb_optimizer == "adagrad" else _capi.OPT_SGD
        c.gemm_mode = gemm_mode
        c.lr, c.eps = cfg.lr, cfg.eps
        c.local_batch, c.rank, c.world = batch, rank, world
        c.comm = comm.h if comm is not None else None
        self._comm = comm
        c.comm_dense = comm_dense.h if comm_dense is not None else None
        self._comm_dense = comm_dense
        if table_owner is not None:
            self._owner = (C.c_int32 * T)(*[int(o) for o in table_owner])
            c.table_owner = self._owner
        c.force_exchange, c.overlap, c.presort, c.wgrad_stream = int(force_exchange), overlap, presort, wgrad_stream
        c.seed, c.init_tables = seed, 1
        c.one_id_per_bag = int(one_id_per_bag)
        c.graph_steps = int(graph_steps)
        self.h = C.c_void_p()
        check(self.L.dlrm_model_create(C.byref(c), C.byref(self.h)))
        info = ModelInfo()
        check(self.L.dlrm_model_info(self.h, C.byref(info)))
        self.info = info
        wdt = torch.bfloat16 if cfg.row_dtype == "bf16" else torch.float32
        self.weight = _view(info.d_weight, (max(info.total_rows, 1), cfg.dim), wdt, self.dev)
        self.params = _view(info.d_params, (info.n_params,), torch.float32, self.dev)
        self.grads = _view(info.d_grads, (info.n_params,), torch.float32, self.dev)
        self.pred = _view(info.d_pred, (batch,), torch.float32, self.dev)
        self.loss = _view(info.d_loss, (1,), torch.float32, self.dev)
        self.err_flag = _view(info.d_error_flag, (1,), torch.int32, self.dev)
        self.opt_state = None if not info.d_opt_state else _view(info.d_opt_state, (max(info.total_rows, 1),),
                                                                 torch.float32, self.dev)
        self.local_tables = []     # (global table, first pool row, rows, first table row)
        for i in range(info.num_local_tables):
            g, f, r, lo = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
            check(self.L.dlrm_model_local_table(self.h, i, C.byref(g), C.byref(f), C.byref(r), C.byref(lo)))
            self.local_tables.append((g.value, f.value, r.value, lo.value))
        order = (C.c_int32 * T)()
        check(self.L.dlrm_model_feature_order(self.h, order))
        self.feature_order = list(order)

    def close(self):
        if getattr(self, "h", None):
            self.L.dlrm_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def layers(self):
        """[(stack, layer, N, K)] bottom layers then top layers."""
        out = []
        for stack, n in ((0, self.info.n_bottom), (1, self.info.n_top)):
            for i in range(n):
                N, K = C.c_int32(), C.c_int32()
                check(self.L.dlrm_model_layer(self.h, stack, i, C.byref(N), C.byref(K), None, None, None))
                out.append((stack, i, N.value, K.value))
        return out

    def get_dense(self):
        Ws, bs = [], []
        for stack, i, N, K in self.layers():
            w, b = np.empty((N, K), np.float32), np.empty(N, np.float32)
            check(self.L.dlrm_model_get_layer(self.h, stack, i, w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
            Ws.append(w)
            bs.append(b)
        return Ws, bs

    def set_dense(self, Ws, bs):
        for (stack, i, N, K), w, b in zip(self.layers(), Ws, bs):
            w, b = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
            assert w.shape == (N, K) and b.shape == (N,)
            check(self.L.dlrm_model_set_layer(self.h, stack, i, w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))

    def init_dense(self, seed):
        check(self.L.dlrm_model_init_dense(self.h, C.c_uint64(seed)))

    def load_batch(self, dense, labels):
        assert dense.is_cuda and dense.stride(1) == 1 and labels.is_contiguous()
        check(self.L.dlrm_model_load_batch(self.h, C.c_void_p(dense.data_ptr()), C.c_int64(dense.stride(0)),
                                           C.c_void_p(labels.data_ptr()), self._stream()))

    TIME_CLASSES = ("linear_forward", "linear_backward_data", "linear_backward_weight", "linear_forward_skinny",
                    "linear_backward_data_skinny", "linear_backward_weight_skinny", "embbag_forward",
                    "embbag_backward_prepare", "embbag_backward_update", "interact_dot_forward", "interact_dot_backward",
                    "comm_forward_exchange", "comm_backward_exchange", "comm_all_reduce", "linear_backward_weight_reduce")

    def enable_timing(self, classes, max_records):
        """classes: names out of TIME_CLASSES (or "all"); [] switches the timing off."""
        names = self.TIME_CLASSES if classes == "all" else classes
        mask = 0
        for n in names:
            mask |= 1 << self.TIME_CLASSES.index(n)
        check(self.L.dlrm_model_enable_timing(self.h, C.c_uint32(mask), C.c_int32(max_records)))

    def collect_timing(self):
        """{class: (milliseconds summed, launches)} since the last collect (synchronises)."""
        n = len(self.TIME_CLASSES)
        ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
        check(self.L.dlrm_model_collect_timing(self.h, ms, cnt))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(self.TIME_CLASSES) if cnt[i]}

    def train_step(self, indices, offsets):
        check(self.L.dlrm_model_train_step(self.h, C.c_void_p(indices.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                           C.c_int64(indices.numel()), self._stream()))
        return self.loss

    def forward(self, indices, offsets):
        check(self.L.dlrm_model_forward(self.h, C.c_void_p(indices.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                        C.c_int64(indices.numel()), self._stream()))
        return self.pred
