"""DLRM train / inference step on one MI355X or table-wise sharded over a node.

Host-side mirror of the reference's model interface (dlrm_libtorch/src/dlrmtest.cpp):
  struct emb   (:144-210)  -> DLRM.embedding_forward / embedding_backward_update
  struct Net2  (:217-278)  -> DLRM.bottom / DLRM.top  (MLPStack)
  main loop    (:473-545)  -> DLRM.train_step
Everything numeric runs in the HIP kernels of libdlrm_amd.so through the C ABI
(dlrm_amd.ops); torch provides device buffers, streams and, when the model is
sharded, the RCCL collectives (torch.distributed, backend "nccl").

Sharding (world > 1): every rank looks its tables up for the GLOBAL batch.
Whole tables (sharding.plan_table_wise): pooled vectors go to the rank that owns
each batch slice with one all-to-all, their gradients come back with another.
Row-wise split tables (sharding.plan_hybrid): every rank pools the rows of its
range, a reduce-scatter over the batch adds the partial sums, an all-gather
brings the gradients of the global batch back and every rank updates its rows.
The MLPs are data parallel and their gradients are summed with one all-reduce
of a single flat buffer.  Losses are means over the global batch, so per-rank
gradients simply add.
"""
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _capi, ops
from .sharding import ShardingPlan, plan_table_wise

# Criteo Terabyte cardinalities with the MLPerf 40M row cap (public MLPerf DLRM spec)
MLPERF_TABLE_ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10,
                     2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
# Criteo Kaggle (Display Advertising Challenge) cardinalities
KAGGLE_TABLE_ROWS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27,
                     14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


@dataclass
class DLRMConfig:
    table_rows: List[int]
    dim: int
    dense_in: int
    bottom: List[int]             # bottom MLP layer widths (every layer ReLU)
    top: List[int]                # top MLP layer widths, last = 1
    interaction: str = "dot"      # "dot" | "cat" (the reference concatenates, src/dlrmtest.cpp:265)
    loss: str = "bce"             # "bce" | "mse" (reference: mse, :522)
    final_relu: bool = False      # reference applies ReLU to the last layer too (:270)
    pooling: str = "sum"          # "sum" | "mean"
    lookups: int = 1              # rows pooled per bag in the synthetic batches
    row_dtype: str = "f32"        # "f32" | "bf16"
    emb_optimizer: str = "sgd"    # "sgd" | "adagrad"
    lr: float = 0.01
    eps: float = 1e-8
    name: str = "custom"

    @staticmethod
    def mlperf(**kw):
        return DLRMConfig(table_rows=list(MLPERF_TABLE_ROWS), dim=128, dense_in=13, bottom=[512, 256, 128],
                          top=[1024, 1024, 512, 256, 1], name="mlperf-dlrm-criteo-tb-40M", **kw)

    @staticmethod
    def kaggle(**kw):
        return DLRMConfig(table_rows=list(KAGGLE_TABLE_ROWS), dim=16, dense_in=13, bottom=[512, 256, 64, 16],
                          top=[512, 256, 1], name="criteo-kaggle", **kw)

    @staticmethod
    def rmc1(rows=500000, **kw):
        """The reference's default synthetic model (src/dlrmtest.cpp:37-49, :234-242)."""
        return DLRMConfig(table_rows=[rows] * 32, dim=32, dense_in=256, bottom=[128, 32], top=[32, 1],
                          interaction="cat", loss="mse", final_relu=True, pooling="mean", lookups=80,
                          name="rmc1", **kw)

    @property
    def num_tables(self):
        return len(self.table_rows)

    @property
    def num_features(self):
        return self.num_tables + 1

    @property
    def top_in(self):
        if self.interaction == "dot":
            F = self.num_features
            return self.dim + F * (F - 1) // 2
        return self.bottom[-1] + self.num_tables * self.dim

    def flops_per_sample_fwd(self):
        f = 0
        k = self.dense_in
        for n in self.bottom:
            f += 2 * k * n
            k = n
        k = self.top_in
        for n in self.top:
            f += 2 * k * n
            k = n
        if self.interaction == "dot":
            f += 2 * self.num_features ** 2 * self.dim
        return f


def _pad4(n):
    return (n + 3) // 4 * 4


def _padk(n):
    """Leading dimension of a matrix whose columns are a GEMM reduction / output index: wide
    ones are rounded up to whole 128-column tiles so that every tile of the fp32 GEMM can use
    its 16-byte loader (the pad columns are zero and never stored to)."""
    return _pad4(n) if n <= 128 else (n + 127) // 128 * 128


class MLPStack:
    """A stack of Linear+ReLU layers (struct Net2's fc1..fc4, src/dlrmtest.cpp:234-272).
    Weights [N, pad4(K)] and biases live in one flat parameter buffer shared by
    all stacks of a model; gradients in its twin."""

    def __init__(self, sizes, params, grads, offset, last_relu=True):
        self.sizes = list(sizes)
        self.last_relu = last_relu
        self.W, self.b, self.dW, self.db = [], [], [], []
        o = offset
        for K, N in zip(sizes[:-1], sizes[1:]):
            kp = _padk(K)
            self.W.append(params[o:o + N * kp].view(N, kp)[:, :K])
            self.dW.append(grads[o:o + N * kp].view(N, kp)[:, :K])
            o += N * kp
            self.b.append(params[o:o + N])
            self.db.append(grads[o:o + N])
            o += _pad4(N)
        self.end = o

    @staticmethod
    def param_count(sizes):
        return sum(N * _padk(K) + _pad4(N) for K, N in zip(sizes[:-1], sizes[1:]))

    @property
    def n_layers(self):
        return len(self.W)

    def init_(self, gen):
        """Xavier-style normal init (DLRM convention); the reference uses
        unscaled randn weights (src/dlrmtest.cpp:110-113), which tests pass in
        explicitly."""
        for W, b in zip(self.W, self.b):
            N, K = W.shape
            W.copy_((torch.randn(N, K, generator=gen) * (2.0 / (K + N)) ** 0.5).to(W.device))
            b.copy_((torch.randn(N, generator=gen) * (1.0 / N) ** 0.5).to(b.device))

    def forward(self, x, acts):
        """acts[i] receives the output of layer i ([B, N_i] views)."""
        for i in range(self.n_layers):
            relu = self.last_relu or i + 1 < self.n_layers
            ops.linear_forward(x, self.W[i], self.b[i], relu, out=acts[i])
            x = acts[i]
        return x

    def backward(self, x_in, acts, dz_last, dz_bufs, ws, dx_first=None):
        """dz_last: gradient w.r.t. the last layer's pre-activation.  dz_bufs[i]
        ([B, N_i]) receives the pre-activation gradient of layer i for
        i < last.  If dx_first is given it receives d(loss)/d(x_in), unmasked."""
        dz = dz_last
        for i in reversed(range(self.n_layers)):
            x = x_in if i == 0 else acts[i - 1]
            _, _, ws = ops.linear_backward_weight(dz, x, dw=self.dW[i], db=self.db[i], workspace=ws)
            if i > 0:
                dz = ops.linear_backward_data(dz, self.W[i], relu_mask=acts[i - 1], out=dz_bufs[i - 1])
            elif dx_first is not None:
                ops.linear_backward_data(dz, self.W[0], out=dx_first)
        return ws


class DLRM:
    def __init__(self, cfg: DLRMConfig, batch: int, device="cuda", rank=0, world=1, plan: Optional[ShardingPlan] = None,
                 seed=0, process_group=None, init_tables=True, overlap=None, cache_fraction=None,
                 cache_rows_per_page=1, replay=False, presort=None, cache_policy="lru",
                 cache_log_pages=4096, wgrad_stream=None, force_exchange=False):
        """batch: LOCAL batch size (samples whose MLP runs on this rank).
        overlap: run the embedding side (kernels + exchanges) on a second stream beside the bottom
        MLP; default: only when sharded (on one GPU the matrix kernels are power-bound and the
        concurrent HBM traffic buys nothing: measured 0.2 %).
        cache_fraction: keep the tables in pinned host DRAM and only this fraction of their pages in
        an HBM cache (dlrm_amd.cache.EmbeddingCache, the reference's `class cache`); every step looks
        its ids up (misses are filled over the host link), pools and updates rows inside the cache's
        slot pool and reclaims down to the capacity.  `train_step(..., next_indices=...)` looks the next
        batch up on a second stream while this one computes.
        wgrad_stream: run the weight gradients of the MLPs on their own stream beside the chain of data
        gradients (both only need a layer's dz).  Default: when the local batch is small (<= 16384 samples):
        then most layers have fewer tiles than the chip has compute units and the two kernels fill it together;
        at the full MLPerf batch every kernel fills the chip alone.
        cache_policy: "lru" (pages or rows, least recently used: `class cache`) or "vec" (the reference's
        `class cache2`, dlrm_amd.vcache.VectorCache: rows cached one by one, every table lists up to
        cache_fraction x (rows of the largest table) rows oldest request first, evicted rows are repacked
        into new pages of the store; float32 rows, 4096-byte pages; cache_log_pages = pages a table may
        write over the run).
        force_exchange: run the sharded schedule (pack, all-to-all, unpack, reduce-scatter / all-gather of
        row-wise tables, all-reduce of the dense gradients) even when world == 1, on the process group
        given: every collective then runs on device buffers in stream order with one rank."""
        _capi.lib()  # fail loudly if the HIP library is missing
        self.cfg, self.B, self.rank, self.world, self.pg = cfg, batch, rank, world, process_group
        self.sharded = world > 1 or bool(force_exchange)
        self.Bg = batch * world
        self.dev = torch.device(device)
        self.plan = plan or plan_table_wise(cfg.table_rows, world, cfg.dim, 2 if cfg.row_dtype == "bf16" else 4)
        if not self.sharded:
            # one rank holds everything whole, in the plan's feature order
            self.my_tables, self.rw_tables = list(self.plan.feature_order), []
        else:
            self.my_tables, self.rw_tables = self.plan.local_tables(rank), list(self.plan.row_wise)
        self.Tl = len(self.my_tables)          # whole tables of this rank
        self.Trw = len(self.rw_tables)         # tables of which this rank holds a row range
        # tables in the order this rank's indices / offsets must list them
        self.local_table_ids = list(self.my_tables) + list(self.rw_tables)
        T, D, F = cfg.num_tables, cfg.dim, cfg.num_features
        dev = self.dev
        self.pooling = _capi.POOL_MEAN if cfg.pooling == "mean" else _capi.POOL_SUM
        if cfg.interaction == "dot" and cfg.bottom[-1] != D:
            raise ValueError("dot interaction needs bottom[-1] == dim")

        # ---- embedding tables of this rank, one allocation, table-major
        ranges = [(0, cfg.table_rows[t]) for t in self.my_tables] + \
                 [self.plan.row_range(t, rank) for t in self.rw_tables]
        rows = [hi - lo for lo, hi in ranges]
        self.local_rows = rows
        tro = [0]
        for r in rows:
            tro.append(tro[-1] + r)
        self.total_rows = tro[-1]
        self.table_row_offsets = torch.tensor(tro, dtype=torch.int64, device=dev)
        wdtype = torch.bfloat16 if cfg.row_dtype == "bf16" else torch.float32
        self.cache = None
        if cache_fraction is not None:
            if self.rw_tables:
                raise ValueError("the cached mode supports whole tables")
            adagrad = cfg.emb_optimizer == "adagrad"     # the rows' state then travels with them through the cache
            esz = 2 if cfg.row_dtype == "bf16" else 4
            rpp = cache_rows_per_page
            self.cache_policy = cache_policy
            if cache_policy == "vec":
                from .vcache import VectorCache
                rpp = 4096 // (D * esz)
                per_step = self.Bg * cfg.lookups                      # new rows a table can list per step
                cap = max(int(max(rows) * cache_fraction), 8 * rpp)
                # listed rows plus the rows that come along with the pages of one step; evictions of all steps
                slots = sum(min(r, (cap + 2 * per_step) * rpp) for r in rows)
                self.cache = VectorCache(rows, D, cap, slots, log_pages_per_table=cache_log_pages, device=dev,
                                         row_dtype=wdtype, row_state=adagrad)
                if init_tables:
                    self.cache.write_tables(seed, self.local_table_ids)
            elif cache_policy == "lru":
                from .cache import EmbeddingCache
                pages = sum((r + rpp - 1) // rpp for r in rows)
                cache_pages = max(int(pages * cache_fraction), 1)
                # a lookup can bring in one page per id of the global batch on top of the capacity
                headroom = min(pages, self.Tl * self.Bg * cfg.lookups)
                # only updated pages are copied back on eviction (inference costs no device-to-host traffic);
                # every buffer of a lookup / reclaim is allocated now (ragged batches: up to twice the nominal ids)
                self.cache = EmbeddingCache(rows, D, cache_pages, cache_pages + headroom, rows_per_page=rpp,
                                            row_dtype=wdtype, device=dev, dirty_only=True, row_state=adagrad,
                                            max_lookup_ids=2 * max(self.Tl * self.Bg * cfg.lookups, 1))
                # victims travel to the store beside the next batch's misses (both directions of the host link at once)
                self.cache.set_async_writeback(headroom)
                if init_tables:
                    self.cache.write(seed, table_ids=self.local_table_ids)
            else:
                raise ValueError("cache_policy must be 'lru' or 'vec'")
            self.weight = self.cache.pool
            self.table_row_offsets = None          # ids are translated to rows of the shared slot pool
            self._rows = None
            self.lookup_stream = torch.cuda.Stream(device=dev)
            init_tables = False
            assert esz * D * rpp % 16 == 0
        else:
            self.weight = torch.empty((max(self.total_rows, 1), D), dtype=wdtype, device=dev)
        if init_tables and self.total_rows:
            # every table is its own stream of uniform [0,1) values, keyed by the GLOBAL
            # table id, so a sharded model holds the same numbers as a single-GPU one
            for lt, t in enumerate(self.local_table_ids):
                ops.table_init_uniform_(self.weight[tro[lt]:tro[lt + 1]], seed=(seed << 20) + 7919 * (t + 1),
                                        first=ranges[lt][0] * D)
        self.opt_state = None
        if cfg.emb_optimizer == "adagrad":
            self.opt_state = (self.cache.state_pool if self.cache is not None else
                              torch.zeros(max(self.total_rows, 1), dtype=torch.float32, device=dev))
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=dev)

        # ---- dense parameters: one flat buffer (+ twin for gradients)
        bsz = [cfg.dense_in] + list(cfg.bottom)
        tsz = [cfg.top_in] + list(cfg.top)
        n_par = MLPStack.param_count(bsz) + MLPStack.param_count(tsz)
        self.params = torch.zeros(n_par, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n_par, dtype=torch.float32, device=dev)
        self.bottom = MLPStack(bsz, self.params, self.grads, 0, last_relu=True)
        self.top = MLPStack(tsz, self.params, self.grads, self.bottom.end, last_relu=cfg.final_relu)
        gen = torch.Generator().manual_seed(seed)  # same on every rank: replicas start identical
        self.bottom.init_(gen)
        self.top.init_(gen)

        # ---- activations
        B = batch
        f32 = dict(dtype=torch.float32, device=dev)
        self.x_dense = torch.zeros((B, _pad4(cfg.dense_in)), **f32)
        self.labels = torch.zeros(B, **f32)
        if cfg.interaction == "dot":
            self.feat = torch.zeros((B, F * D), **f32)            # [B, F, D]; feature 0 = bottom output
            self.dfeat = torch.zeros((B, F * D), **f32)
            self.inter_ld = _padk(cfg.top_in)
            self.inter = torch.zeros((B, self.inter_ld), **f32)
            self.dinter = torch.zeros((B, self.inter_ld), **f32)
            self.emb_col0 = D
        else:
            self.emb_col0 = cfg.bottom[-1]
            self.inter_ld = _padk(cfg.top_in)
            self.feat = torch.zeros((B, self.inter_ld), **f32)    # [bottom out | pooled tables], read by the top MLP
            self.dfeat = torch.zeros((B, self.inter_ld), **f32)
            self.inter, self.dinter = self.feat, self.dfeat
        self.bot_acts = [torch.zeros((B, _padk(n)), **f32)[:, :n] for n in cfg.bottom[:-1]]
        self.bot_acts.append(self.feat[:, :cfg.bottom[-1]])
        self.bot_dz = [torch.zeros((B, _padk(n)), **f32)[:, :n] for n in cfg.bottom[:-1]]
        self.top_acts = [torch.zeros((B, _padk(n)), **f32)[:, :n] for n in cfg.top[:-1]]
        self.pred = torch.zeros((B, 1), **f32)
        self.top_acts.append(self.pred)
        self.top_dz = [torch.zeros((B, _padk(n)), **f32)[:, :n] for n in cfg.top[:-1]]
        self.dpred = torch.zeros((B, 1), **f32)
        self.loss = torch.zeros(1, **f32)
        self.ws_dense = None
        self.ws_sparse = None
        # The sparse update's sort needs the ids only: it is queued on its own stream at the start of the
        # step and runs beside the forward pass (presort; on by default on the GPU).
        self.presort = (self.dev.type == "cuda") if presort is None else bool(presort)
        self.prep = torch.cuda.Stream(device=dev) if (self.presort and self.dev.type == "cuda") else None
        self.ws_sparse_rw = None
        self.replay = bool(replay) and self.dev.type == "cuda"
        self._plan, self._plan_items, self._plan_key = None, None, None
        self.plan_timing_runs = 0       # > 0: the recorded step times every entry (bench.py)
        # The EmbeddingBag kernels (and, sharded, their exchanges) are HBM / link bound, the bottom MLP
        # is matrix-pipe bound and the two touch different columns of `feat` / `dfeat`: they can run
        # concurrently, the embedding side on a second HIP stream.
        if overlap is None:
            overlap = self.sharded
        self.side = torch.cuda.Stream(device=dev) if (overlap and self.dev.type == "cuda") else None
        if wgrad_stream is None:
            wgrad_stream = batch <= 32768
        self.wgrad = torch.cuda.Stream(device=dev) if (wgrad_stream and self.dev.type == "cuda") else None

        # ---- exchange buffers (sharded only)
        if self.sharded:
            self.send_splits = [B * self.Tl * D] * world
            self.recv_splits = [B * len(self.plan.tables_of_rank[j]) * D for j in range(world)]
            self.a2a_send = torch.zeros(max(self.Bg * self.Tl * D, 1), **f32)   # [Bg, Tl*D]
            n_tw = self.plan.n_table_wise
            self.a2a_recv = torch.zeros(max(B * n_tw * D, 1), **f32)            # blocks [B, Tj*D], j = 0..W-1
            self.g_send = torch.zeros(max(B * n_tw * D, 1), **f32)
            self.g_recv = torch.zeros(max(self.Bg * self.Tl * D, 1), **f32)
            if self.Trw:
                self.shard_row_begin = torch.tensor([lo for lo, _ in ranges[self.Tl:]], dtype=torch.int64, device=dev)
                self.rw_part = torch.zeros((self.Bg, self.Trw * D), **f32)   # partial sums / gathered gradients
                self.rw_local = torch.zeros((B, self.Trw * D), **f32)       # this rank's batch slice

    # ------------------------------------------------------------ embeddings

    def _emb_kwargs(self, indices, offsets):
        """Descriptor arguments for the whole tables of this rank: the first Tl tables of the
        local order (offsets are table-major and absolute, so a prefix is a valid CSR)."""
        n = self.Tl * self.Bg
        tro = None if self.table_row_offsets is None else self.table_row_offsets[:self.Tl + 1]
        return dict(weight=self.weight, indices=indices, offsets=offsets[:n + 1], batch=self.Bg, num_tables=self.Tl,
                    table_row_offsets=tro, pooling=self.pooling, include_last_offset=True, error_flag=self.err_flag)

    # ---------------------------------------------------------- cached tables

    def _table_pos(self, indices, table_pos):
        if table_pos is not None:
            return [int(x) for x in table_pos]
        n = indices.numel()
        if n % max(self.Tl, 1):
            raise ValueError("ragged batches need table_pos (where each table's ids start)")
        return [t * (n // self.Tl) for t in range(self.Tl + 1)]

    def lookup(self, indices, table_pos=None, train=True):
        """create_emb's gather (src/dlrmtest.cpp:53-86): slot-pool row of every id of the batch.  train: the
        rows will be updated, so their pages must be copied back when they leave the cache."""
        rows = self.cache.get_emb_vectors(indices, self._table_pos(indices, table_pos))
        if train and self.cache_policy == "lru":
            self.cache.mark_dirty(rows)
        return rows

    def _pipeline_next(self, next_indices, next_table_pos):
        """Cache work that runs beside this step's kernels, on the lookup stream: the second half of the
        previous step's reclaim, the lookup of the NEXT batch (its misses cross the host link while this
        batch computes), then -- once this step's row update is done -- the first half of this step's
        reclaim.  The host only waits for a read-back that finished a step ago."""
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.lookup_stream):
            self.cache.reclaim_finish()
            rows = self.lookup(next_indices, next_table_pos)
            ev = torch.cuda.Event()
            ev.record()
            self._rows = (next_indices, rows, ev)
            self.lookup_stream.wait_stream(main)        # this step's rows are updated
            # the reference reclaims BEFORE it looks the next batch up (src/dlrmtest.cpp:484-494); here the
            # order is reversed, so the pages of the batch just looked up (age 0) must not be victims: the
            # next step pools from them while the second half of this reclaim runs
            self.cache.set_protect_age(1)
            self.cache.reclaim_begin()

    def _rw_kwargs(self, indices, offsets):
        """Descriptor arguments for the row ranges this rank holds of the row-wise tables."""
        n = self.Tl * self.Bg
        return dict(weight=self.weight, indices=indices, offsets=offsets[n:], batch=self.Bg, num_tables=self.Trw,
                    table_row_offsets=self.table_row_offsets[self.Tl:], pooling=self.pooling,
                    include_last_offset=True, error_flag=self.err_flag, shard_row_begin=self.shard_row_begin)

    # Collectives.  RCCL works on device buffers in stream order.  gloo is the rehearsal backend (CPU ranks,
    # or several ranks sharing one GPU): device buffers are staged through host memory there.
    def _staged(self):
        import torch.distributed as dist
        return self.dev.type == "cuda" and dist.get_backend(self.pg) == "gloo"

    def _all_to_all(self, out, inp, out_splits, in_splits):
        import torch.distributed as dist
        if self._staged():
            h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, h_in, out_splits, in_splits, group=self.pg)
            out.copy_(h_out)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.pg)

    def _all_reduce(self, t):
        import torch.distributed as dist
        if self._staged():
            h = t.cpu()
            dist.all_reduce(h, group=self.pg)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.pg)

    def _all_gather(self, out, inp):
        import torch.distributed as dist
        if self._staged():
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h_out, inp.cpu(), group=self.pg)
            out.copy_(h_out)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.pg)

    def _reduce_scatter_batch(self, out, inp):
        """out [B, C] = sum over ranks of their inp [Bg, C] rows of this rank's batch slice."""
        import torch.distributed as dist
        if dist.get_backend(self.pg) == "gloo":       # no reduce-scatter there
            self._all_reduce(inp)
            out.copy_(inp[self.rank * self.B:(self.rank + 1) * self.B])
        else:
            dist.reduce_scatter_tensor(out, inp, group=self.pg)

    def exchange_ids(self, local_indices, hotness=None):
        """The index exchange of a sharded run.  local_indices: ids of THIS rank's batch slice for all tables,
        table-major in the plan's feature order (bag (t', b) holds hotness[feature_order[t']] ids; default
        cfg.lookups each) -- what CriteoBinReader(table_order=plan.feature_order, rank=..., world=...) returns.
        Every rank needs the ids of ITS tables for the GLOBAL batch: the blocks of whole tables travel with one
        all-to-all (block p of local_indices is exactly what rank p owns), those of row-wise split tables with an
        all-gather; a regroup kernel lays them out table-major over the global batch.  Returns (indices, offsets)
        as train_step / forward take them.  The reference is one process and slices one tensor
        (src/dlrmtest.cpp:53-63, :150)."""
        cfg, B, W = self.cfg, self.B, self.world
        T = cfg.num_tables
        hot = [cfg.lookups] * T if hotness is None else [int(h) for h in hotness]
        order = self.plan.feature_order
        dev, dt = local_indices.device, local_indices.dtype
        per_rank = [sum(hot[t] for t in self.plan.tables_of_rank[p]) for p in range(self.plan.world)] \
            if self.sharded else [sum(hot[t] for t in order)]
        n_tw = sum(per_rank)
        s_rw = sum(hot[t] for t in self.rw_tables)
        assert local_indices.numel() == B * (n_tw + s_rw), "local_indices does not match batch x hotness"
        mine = per_rank[self.rank] if self.sharded else n_tw

        def hot_off(tables):
            ho = [0]
            for t in tables:
                ho.append(ho[-1] + hot[t])
            return torch.tensor(ho, dtype=torch.int32, device=dev)
        Bg = self.Bg
        indices = torch.empty(max(Bg * (mine + s_rw), 1), dtype=dt, device=dev)
        offsets = torch.empty((self.Tl + self.Trw) * Bg + 1, dtype=dt, device=dev)
        if not self.sharded:
            ops.ids_regroup(local_indices, 1, B, self.Tl, hot_off(self.my_tables), 0, indices, offsets)
            return indices[:Bg * mine], offsets
        recv = torch.empty(max(W * B * mine, 1), dtype=dt, device=dev)
        self._all_to_all(recv[:W * B * mine], local_indices[:B * n_tw], [B * mine] * W, [B * p for p in per_rank])
        if self.Tl:
            ops.ids_regroup(recv, W, B, self.Tl, hot_off(self.my_tables), 0, indices, offsets)
        if self.Trw:
            recv_rw = torch.empty(W * B * s_rw, dtype=dt, device=dev)
            self._all_gather(recv_rw, local_indices[B * n_tw:].contiguous())
            ops.ids_regroup(recv_rw, W, B, self.Trw, hot_off(self.rw_tables), Bg * mine, indices[Bg * mine:],
                            offsets[self.Tl * Bg:])
        elif not self.Tl:
            offsets.zero_()
        return indices[:Bg * (mine + s_rw)], offsets

    # The step is written once, as a schedule: a list of (kind, stream, fn) items.
    #   ("ops",  stream, fn)   fn() calls dlrm_amd.ops operators only (launched now, or recorded
    #                          into a dlrm_plan and replayed as one C call)
    #   ("host", stream, fn)   fn() is host work issued on that stream: the collectives
    #   ("wait", a, b)         stream a waits for everything queued on stream b
    # Streams are named "main" (the caller's) and "emb" (the second stream of the embedding side, or
    # the caller's when there is no overlap).

    def _emb_forward_items(self, indices, offsets):
        """emb::forward (src/dlrmtest.cpp:170-206) for this rank's tables over the
        global batch, then the exchange that leaves feat[:, emb_col0:] holding
        the pooled vectors of ALL tables for the local batch."""
        D, B = self.cfg.dim, self.B
        items = []
        if not self.sharded:
            def pool():
                ops.embbag_forward(out=self.feat[:, self.emb_col0:], out_ld=self.feat.stride(0),
                                   **self._emb_kwargs(indices, offsets))
            return [("ops", "emb", pool)]
        import torch.distributed as dist
        n_send, n_recv = self.Bg * self.Tl * D, B * self.plan.n_table_wise * D   # exact sizes (buffers hold >= 1 element)
        if self.Tl:
            items.append(("ops", "emb", lambda: ops.embbag_forward(out=self.a2a_send, out_ld=self.Tl * D,
                                                                   **self._emb_kwargs(indices, offsets))))
        items.append(("host", "emb", lambda: self._all_to_all(
            self.a2a_recv[:n_recv], self.a2a_send[:n_send], self.recv_splits, self.send_splits)))

        def unpack():
            o = 0
            for j in range(self.world):
                first, cnt = self.plan.feature_slot(j)
                if cnt:
                    blk = self.a2a_recv[o:o + B * cnt * D].view(B, cnt * D)
                    c0 = self.emb_col0 + first * D
                    ops.copy_2d_(self.feat[:, c0:c0 + cnt * D], blk)
                o += B * cnt * D
        if self.plan.n_table_wise:
            items.append(("ops", "emb", unpack))
        if self.Trw:
            c0 = self.emb_col0 + self.plan.n_table_wise * D
            items.append(("ops", "emb", lambda: ops.embbag_forward(out=self.rw_part, out_ld=self.Trw * D,
                                                                   **self._rw_kwargs(indices, offsets))))
            items.append(("host", "emb", lambda: self._reduce_scatter_batch(self.rw_local, self.rw_part)))
            items.append(("ops", "emb", lambda: ops.copy_2d_(self.feat[:, c0:c0 + self.Trw * D], self.rw_local)))
        return items

    def _presort_items(self, indices, offsets):
        """First half of the sparse update for this rank's tables (ids only), on the "prep" stream."""
        if not self.presort:
            return []
        items = [("wait", "prep", "main")]

        def prep_tw():
            self.ws_sparse = ops.embbag_backward_prepare(workspace=self.ws_sparse, **self._emb_kwargs(indices, offsets))

        def prep_rw():
            self.ws_sparse_rw = ops.embbag_backward_prepare(workspace=self.ws_sparse_rw,
                                                            **self._rw_kwargs(indices, offsets))
        if self.Tl:
            items.append(("ops", "prep", prep_tw))
        if self.Trw and self.sharded:
            items.append(("ops", "prep", prep_rw))
        return items

    def _emb_backward_items(self, indices, offsets):
        """autograd of embedding_bag + optimizer step + write-back
        (src/dlrmtest.cpp:531-535) fused: gradients of the pooled vectors go
        back to the owning ranks and are applied to the rows."""
        D, B = self.cfg.dim, self.B
        cfg = self.cfg

        def update(kwargs_fn, grad, grad_ld, rw=False):
            kw = dict(lr=cfg.lr, eps=cfg.eps, opt_state=self.opt_state,
                      optimizer=_capi.OPT_ADAGRAD_ROWWISE if cfg.emb_optimizer == "adagrad" else _capi.OPT_SGD)

            def run():
                if self.presort:       # the occurrences were sorted at the start of the step
                    ops.embbag_backward_apply(grad=grad, grad_ld=grad_ld,
                                              workspace=self.ws_sparse_rw if rw else self.ws_sparse,
                                              **kw, **kwargs_fn(indices, offsets))
                else:
                    self.ws_sparse = ops.embbag_backward_update(grad=grad, grad_ld=grad_ld, workspace=self.ws_sparse,
                                                                **kw, **kwargs_fn(indices, offsets))
            return run
        items = [("wait", "emb", "prep")] if self.presort else []
        if not self.sharded:
            return items + [("ops", "emb", update(self._emb_kwargs, self.dfeat[:, self.emb_col0:],
                                                  self.dfeat.stride(0)))]
        import torch.distributed as dist

        def pack():
            o = 0
            for j in range(self.world):
                first, cnt = self.plan.feature_slot(j)
                if cnt:
                    blk = self.g_send[o:o + B * cnt * D].view(B, cnt * D)
                    c0 = self.emb_col0 + first * D
                    ops.copy_2d_(blk, self.dfeat[:, c0:c0 + cnt * D])
                o += B * cnt * D
        n_send, n_recv = self.Bg * self.Tl * D, B * self.plan.n_table_wise * D
        if self.plan.n_table_wise:
            items.append(("ops", "emb", pack))
        items.append(("host", "emb", lambda: self._all_to_all(
            self.g_recv[:n_send], self.g_send[:n_recv], self.send_splits, self.recv_splits)))
        if self.Tl:
            items.append(("ops", "emb", update(self._emb_kwargs, self.g_recv, self.Tl * D)))
        if self.Trw:
            c0 = self.emb_col0 + self.plan.n_table_wise * D
            items.append(("ops", "emb", lambda: ops.copy_2d_(self.rw_local, self.dfeat[:, c0:c0 + self.Trw * D])))
            items.append(("host", "emb", lambda: self._all_gather(self.rw_part, self.rw_local)))
            items.append(("ops", "emb", update(self._rw_kwargs, self.rw_part, self.Trw * D, rw=True)))
        return items

    def _forward_items(self, indices, offsets, train=False):
        """Net2::forward (src/dlrmtest.cpp:254-272) around emb::forward."""
        cfg = self.cfg
        items = self._presort_items(indices, offsets) if train else []
        items.append(("wait", "emb", "main"))    # earlier readers of feat and the input copies are done
        items += self._emb_forward_items(indices, offsets)
        items.append(("ops", "main", lambda: self.bottom.forward(self.x_dense[:, :cfg.dense_in], self.bot_acts)))
        items.append(("wait", "main", "emb"))

        def top():
            if cfg.interaction == "dot":
                ops.interact_dot_forward(self.feat.view(self.B, cfg.num_features, cfg.dim), out=self.inter,
                                         out_ld=self.inter_ld)
            self.top.forward(self.inter[:, :cfg.top_in], self.top_acts)
        items.append(("ops", "main", top))
        return items

    def _mlp_backward_items(self, stack, x_in, acts, dz_last, dz_bufs, dx_first=None):
        """MLPStack.backward as schedule items.  With a weight-gradient stream every layer's dW / db kernels go
        there (they wait for the layer's dz) while the data gradients continue on the caller's stream."""
        if self.wgrad is None:
            def whole():
                self.ws_dense = stack.backward(x_in, acts, dz_last, dz_bufs, self.ws_dense, dx_first=dx_first)
            return [("ops", "main", whole)]
        items = []
        for i in reversed(range(stack.n_layers)):
            dz = dz_last if i == stack.n_layers - 1 else dz_bufs[i]
            x = x_in if i == 0 else acts[i - 1]

            def wgrad(dz=dz, x=x, i=i):
                _, _, self.ws_dense = ops.linear_backward_weight(dz, x, dw=stack.dW[i], db=stack.db[i],
                                                                 workspace=self.ws_dense)
            items.append(("wait", "wgrad", "main"))          # dz of this layer is complete
            items.append(("ops", "wgrad", wgrad))
            if i > 0:
                items.append(("ops", "main", lambda dz=dz, i=i: ops.linear_backward_data(
                    dz, stack.W[i], relu_mask=acts[i - 1], out=dz_bufs[i - 1])))
            elif dx_first is not None:
                items.append(("ops", "main", lambda dz=dz: ops.linear_backward_data(dz, stack.W[0], out=dx_first)))
        return items

    def _train_items(self, indices, offsets):
        """One iteration of the reference's loop body (src/dlrmtest.cpp:505-535)."""
        cfg = self.cfg
        items = self._forward_items(indices, offsets, train=True)

        def loss_fn():
            if cfg.loss == "bce":
                ops.loss_bce_logits(self.pred.view(-1), self.labels, loss=self.loss, dlogit=self.dpred.view(-1),
                                    mean_over=self.Bg)
            else:
                ops.loss_mse(self.pred.view(-1), self.labels, relu_mask_pred=cfg.final_relu, loss=self.loss,
                             dpred=self.dpred.view(-1), mean_over=self.Bg)
        items.append(("ops", "main", loss_fn))
        if cfg.interaction == "dot":
            items += self._mlp_backward_items(self.top, self.inter[:, :cfg.top_in], self.top_acts, self.dpred, self.top_dz,
                                              dx_first=self.dinter[:, :cfg.top_in])
            items.append(("ops", "main", lambda: ops.interact_dot_backward(
                self.feat.view(self.B, cfg.num_features, cfg.dim), self.dinter[:, :cfg.top_in], relu_mask_feature0=True,
                out=self.dfeat.view(self.B, cfg.num_features, cfg.dim))))
        else:
            items += self._mlp_backward_items(self.top, self.inter[:, :cfg.top_in], self.top_acts, self.dpred, self.top_dz)

            def concat_backward():
                nb = cfg.bottom[-1]
                dz0 = self.top_dz[0] if self.top.n_layers > 1 else self.dpred
                # bottom part of the concatenation is a ReLU output, the pooled part is not
                ops.linear_backward_data(dz0, self.top.W[0][:, :nb], relu_mask=self.feat[:, :nb], out=self.dfeat[:, :nb])
                ops.linear_backward_data(dz0, self.top.W[0][:, nb:], out=self.dfeat[:, nb:cfg.top_in])
            items.append(("ops", "main", concat_backward))
        # sparse rows (exchange + fused update) beside the bottom MLP's backward
        items.append(("wait", "emb", "main"))     # dfeat is complete
        items += self._emb_backward_items(indices, offsets)

        items += self._mlp_backward_items(self.bottom, self.x_dense[:, :cfg.dense_in], self.bot_acts,
                                          self.dfeat[:, :cfg.bottom[-1]], self.bot_dz)
        items.append(("wait", "main", "wgrad"))   # every dense gradient is there
        if self.sharded:
            # every dense gradient is complete: the all-reduce runs beside the sparse row update of the other stream
            items.append(("host", "main", lambda: self._all_reduce(self.grads)))
        items.append(("wait", "main", "emb"))     # rows are updated before anyone reuses dfeat
        items.append(("ops", "main", lambda: ops.sgd_step_(self.params, self.grads, cfg.lr)))
        return items

    def _streams(self):
        main = torch.cuda.current_stream() if self.dev.type == "cuda" else None
        return {"main": main, "emb": self.side if self.side is not None else main,
                "prep": self.prep if self.prep is not None else main,
                "wgrad": self.wgrad if self.wgrad is not None else main}

    def _run_items(self, items, plan=None):
        """Executes a schedule.  Without a plan the "ops" items launch their operators now; with one,
        every "ops" item is a pair (first, last) of recorded entries replayed by one C call."""
        st = self._streams()
        trace = os.environ.get("DLRM_TRACE_ITEMS")      # debugging aid: synchronise and report after every item
        for n, (kind, a, b) in enumerate(items):
            if trace:
                if n:
                    torch.cuda.synchronize()
                print("[rank %d] item %d/%d %s %s" % (self.rank, n, len(items), kind, a), file=sys.stderr, flush=True)
            if kind == "wait":
                if st[a] is not None and st[a] != st[b]:
                    st[a].wait_stream(st[b])
            elif kind == "ops" and plan is not None:
                plan.run(b[0], b[1], st[a])
            elif st[a] is None or st[a] == st["main"]:
                b()
            else:
                with torch.cuda.stream(st[a]):
                    b()

    def _record(self, items, indices, offsets):
        """Traces the "ops" items of a schedule into a dlrm_plan (nothing is launched)."""
        from .plan import StepRecorder
        rec = StepRecorder(indices, offsets)
        out = []
        with rec:
            for kind, a, b in items:
                if kind == "ops":
                    first = rec.size
                    b()
                    out.append((kind, a, (first, rec.size)))
                else:
                    out.append((kind, a, b))
        return rec.plan, out

    # ------------------------------------------------------------------ step

    def load_batch(self, dense, labels):
        self.x_dense[:, :self.cfg.dense_in].copy_(dense)
        self.labels.copy_(labels.reshape(-1))

    def forward(self, indices, offsets, table_pos=None):
        """Inference: Net2::forward (src/dlrmtest.cpp:254-272) around emb::forward.  Cached tables (page-LRU
        policy): ids are looked up first, nothing is marked for write-back, the cache is reclaimed afterwards."""
        if self.cache is not None:
            if self.cache_policy != "lru":
                raise ValueError("inference through the cache is offered for the page-LRU policy")
            main = torch.cuda.current_stream()
            self.lookup_stream.wait_stream(main)
            with torch.cuda.stream(self.lookup_stream):
                self.cache.reclaim_finish()
                rows = self.lookup(indices, table_pos, train=False)
            main.wait_stream(self.lookup_stream)
            rows.record_stream(main)
            self._rows = None
            self._run_items(self._forward_items(rows, offsets))
            self._reclaim_now()
            return self.pred
        self._run_items(self._forward_items(indices, offsets))
        return self.pred

    def train_step(self, indices, offsets, next_indices=None, table_pos=None, next_table_pos=None):
        """One iteration of the reference's loop body (src/dlrmtest.cpp:505-535):
        forward, loss, backward, optimizer step on every parameter, sparse
        row update.  Returns the device scalar holding this rank's share of the
        global-mean loss.  Cached tables: ids are looked up first (or were prefetched by the
        previous call through next_indices) and the cache is reclaimed at the end
        (reclaim_emb_cache, src/dlrmtest.cpp:397-412, :494).
        With `replay` the step's operators are recorded into a dlrm_plan on the first call and
        replayed afterwards (same kernels, same order, one C call per stretch between collectives)."""
        pipelined = False
        ids = indices
        if self.cache is not None and self.cache_policy == "vec":
            # cache2's order of calls: lookups of every table, (the step), reclaim, save_emb's restore
            if next_indices is not None:
                raise ValueError("the vector-granular cache does not look ahead")
            self._vec_pos = self._table_pos(indices, table_pos)
            indices = self.cache.get_emb_vectors(indices, self._vec_pos)
        elif self.cache is not None:
            if self._rows is not None and self._rows[0] is indices:
                _, rows, ev = self._rows
                torch.cuda.current_stream().wait_event(ev)
                rows.record_stream(torch.cuda.current_stream())   # allocated on the lookup stream
            else:
                main = torch.cuda.current_stream()
                self.lookup_stream.wait_stream(main)
                with torch.cuda.stream(self.lookup_stream):       # not looked up ahead: do it now
                    self.cache.reclaim_finish()
                    rows = self.lookup(indices, table_pos)
                main.wait_stream(self.lookup_stream)
                rows.record_stream(main)
            self._rows = None
            pipelined = next_indices is not None
            indices = rows
        if not self.replay:
            self._run_items(self._train_items(indices, offsets))
        else:
            key = (indices.dtype, indices.numel(), offsets.numel())
            if self._plan is None or self._plan_key != key:
                self._plan, self._plan_items = self._record(self._train_items(indices, offsets), indices, offsets)
                self._plan_key = key
                if self.plan_timing_runs:
                    self._plan.enable_timing(self.plan_timing_runs)
            self._plan.set_batch(indices, offsets)
            self._run_items(self._plan_items, self._plan)
        if self.cache is not None and self.cache_policy == "vec":
            self.cache.reclaim()
            self.cache.restore(ids, self._vec_pos)
        elif self.cache is not None:
            if pipelined:
                self._pipeline_next(next_indices, next_table_pos)
            else:
                self._reclaim_now()
        return self.loss

    def _reclaim_now(self):
        """reclaim_emb_cache in the reference's place: after the batch's rows are done with, any page may go."""
        self.lookup_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.lookup_stream):
            self.cache.set_protect_age(0)
            self.cache.reclaim()
        torch.cuda.current_stream().wait_stream(self.lookup_stream)
