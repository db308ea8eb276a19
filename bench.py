#!/usr/bin/env python3
"""Headline benchmark: MLPerf-DLRM-shaped training step, samples/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: the whole model (26 tables capped at 40M rows, dim 128, ~96 GB fp32) on
one MI355X, batch 65536.  N > 1 (launched with torch.distributed.run, one rank
per GPU): the same GLOBAL batch 65536, tables sharded table-wise, pooled
embeddings exchanged with RCCL all-to-all, MLP gradients all-reduced -- strong
scaling.  A step = forward, loss, backward, fused sparse row update and dense
SGD step on synthetic data already resident in HBM.

Rank 0 prints one JSON line.

Host of the step (--host): "c" (default) = the step lives in libdlrm_amd.so behind dlrm_model_* (placement,
streams, exchanges over dlrm_comm / RCCL, operator order: csrc/model.hip), this script only feeds batches -- what
a C++ host runs (adapter/dlrm_train.cpp is the same loop without Python).  `roofline` then describes the operator
class that takes the most time per step, timed with HIP events around its launches INSIDE the timed steps (on
the streams they run on); `kernels` lists every class, timed the same way in a few extra steps after the timed
region (events around all ~60 launches of a step cost throughput, so they stay out of the timed steps).
"python" = dlrm_amd.model (the recorded step replayed through dlrm_plan_run, or --eager); needed for the cached
modes; every class is timed inside the timed steps there.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12       # B/s, MI355X spec
F32_MFMA_PEAK = 157.3e12  # FLOP/s, v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK = 2.5e15   # FLOP/s, v_mfma_f32_32x32x16_bf16 dense peak
# A float32 product evaluated as six bf16 MFMA terms (split of each operand into three bf16
# pieces) can at best run at a sixth of the bf16 rate: that is the roofline of that kernel.
BF16X3_PEAK = BF16_MFMA_PEAK / 6.0
# ... and as three f16 MFMA terms (two f16 pieces per operand; the f16 pipe has the bf16 rate) at a third of it
F16X2_PEAK = BF16_MFMA_PEAK / 3.0

# evaluation of the big layers' float32 products: name -> (C ABI gemm_mode, kernel label, peak, MFMA terms, notes)
GEMM_CHOICES = {
    "f32": (1, "gemm_f32_kernel", F32_MFMA_PEAK, 1.0, "float32 MFMA dense peak", "float32 MFMA"),
    "bf16x3": (2, "gemm_split_kernel<3 bf16 pieces>", BF16X3_PEAK, 6.0,
               "float32 product = 6 bf16 MFMA terms: peak = 2500 TFLOP/s bf16 dense / 6",
               "float32 in/out; products as 6 bf16 MFMA terms of 3-way bf16 splits, f32 accumulate"),
    "f16x2": (3, "gemm_f16q_kernel (v_mfma_f32_16x16x32_f16, weights pre-split; weight gradient: gemm_split_kernel<2 f16 pieces>)",
              F16X2_PEAK, 3.0,
              "float32 product = 3 f16 MFMA terms: peak = 2500 TFLOP/s f16 dense / 3",
              "float32 in/out; products as 3 f16 MFMA terms of 2-way f16 splits of the operands scaled by powers of two, "
              "f32 accumulate (representation error about one float32 ulp per product)"),
}


def gemm_choice(args, local_batch):
    """--gemm auto: f16 pieces for local batches of 8192 and above (what dlrm_model_* does for DLRM_GEMM_DEFAULT),
    bf16 pieces below; DLRM_GEMM_MODE=0 / 2 in the environment force the float32 MFMA / the f16 pieces."""
    from dlrm_amd import ops
    name = args.gemm
    if name == "auto":
        name = {0: "f32", 2: "f16x2"}.get(ops.get_gemm_mode(), "f16x2" if local_batch >= 8192 else "bf16x3")
    return name


class KernelTimer:
    """Wraps dlrm_amd.ops entry points with HIP event pairs on the launch
    stream and accumulates algorithmic work per kernel class."""

    def __init__(self):
        self.records = []   # (name, ev0, ev1, work, unit_kind)
        self.enabled = False

    def wrap(self, ops, name, work_fn, kind):
        orig = getattr(ops, name)
        timer = self

        def wrapped(*a, **kw):
            if not timer.enabled:
                return orig(*a, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **kw)
            e1.record()
            key, work, knd = classify(name, a, kw, work_fn(*a, **kw), kind)
            timer.records.append((key, e0, e1, work, knd))
            return r

        setattr(ops, name, wrapped)

    def summary(self):
        out = {}
        for name, e0, e1, work, kind in self.records:
            d = out.setdefault(name, {"ms": 0.0, "work": 0.0, "launches": 0, "kind": kind})
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
            d["launches"] += 1
        return out


def classify(name, a, kw, work, kind):
    """Layers that are no matrix product (13 input columns, one output column) run in the one-pass kernels of
    skinny.hip: they are listed on their own, priced by the bytes of their large operands."""
    if name == "linear_forward":
        M, (N, K) = a[0].shape[0], a[1].shape
        if (K <= 16 and N >= 64) or N == 1:
            return name + "_skinny", 4.0 * M * (N + K), "hbm"
    elif name == "linear_backward_data":
        M, (N, K) = a[0].shape[0], a[1].shape
        if N == 1:
            masked = len(a) > 2 and a[2] is not None or kw.get("relu_mask") is not None
            return name + "_skinny", 4.0 * M * (N + K * (2 if masked else 1)), "hbm"
    elif name == "linear_backward_weight":
        M, N, K = a[0].shape[0], a[0].shape[1], a[1].shape[1]
        if (K <= 16 and N >= 64) or N == 1:
            return name + "_skinny", 4.0 * M * (N + K), "hbm"
    return name, work, kind


def install_timers(ops, model_ref):
    t = KernelTimer()

    def gemm_flops_fwd(x, w, bias, relu, out=None):
        return 2.0 * x.shape[0] * w.shape[0] * w.shape[1]

    def gemm_flops_bwd_data(dz, w, relu_mask=None, out=None):
        return 2.0 * dz.shape[0] * w.shape[0] * w.shape[1]

    def gemm_flops_bwd_weight(dz, x, dw=None, db=None, workspace=None, want_db=True):
        return 2.0 * dz.shape[0] * dz.shape[1] * x.shape[1]

    # Algorithmic bytes of the EmbeddingBag kernels.  A call covers `num_tables` of the rank's tables
    # (the whole ones, or its row ranges of the row-wise split ones: shard_row_begin given); of a
    # row-wise table only the ids inside the rank's range are gathered / updated (about 1/world).
    def emb_fwd_bytes(weight=None, indices=None, offsets=None, batch=None, num_tables=1, shard_row_begin=None, **kw):
        D = weight.shape[1]
        n_idx = batch * num_tables * model_ref["lookups"]
        frac = 1.0 / model_ref["world"] if shard_row_begin is not None else 1.0
        return (frac * n_idx * D * weight.element_size() + batch * num_tables * D * 4 +
                n_idx * indices.element_size() + offsets.numel() * offsets.element_size())

    def emb_bwd_bytes(weight=None, indices=None, offsets=None, batch=None, num_tables=1, shard_row_begin=None, **kw):
        D = weight.shape[1]
        n_idx = batch * num_tables * model_ref["lookups"]
        U = model_ref["unique_rows_rw" if shard_row_begin is not None else "unique_rows"]
        return batch * num_tables * D * 4 + 2 * U * D * weight.element_size() + n_idx * indices.element_size()

    def inter_fwd_bytes(x, out=None, out_ld=None):
        B, F, D = x.shape
        return B * (F * D + D + F * (F - 1) // 2) * 4

    def inter_bwd_bytes(x, dout, relu_mask_feature0=False, out=None):
        B, F, D = x.shape
        return B * (2 * F * D + D + F * (F - 1) // 2) * 4

    t.work = {"linear_forward": (gemm_flops_fwd, "mfma"), "linear_backward_data": (gemm_flops_bwd_data, "mfma"),
              "linear_backward_weight": (gemm_flops_bwd_weight, "mfma"), "embbag_forward": (emb_fwd_bytes, "hbm"),
              "embbag_backward_update": (emb_bwd_bytes, "hbm"), "embbag_backward_apply": (emb_bwd_bytes, "hbm"),
              "embbag_backward_prepare": (lambda *a, **kw: 0.0, "hbm"), "interact_dot_forward": (inter_fwd_bytes, "hbm"),
              "interact_dot_backward": (inter_bwd_bytes, "hbm")}
    t.wrap(ops, "linear_forward", gemm_flops_fwd, "mfma")
    t.wrap(ops, "linear_backward_data", gemm_flops_bwd_data, "mfma")
    t.wrap(ops, "linear_backward_weight", gemm_flops_bwd_weight, "mfma")
    t.wrap(ops, "embbag_forward", emb_fwd_bytes, "hbm")
    t.wrap(ops, "embbag_backward_update", emb_bwd_bytes, "hbm")
    t.wrap(ops, "embbag_backward_apply", emb_bwd_bytes, "hbm")
    t.wrap(ops, "interact_dot_forward", inter_fwd_bytes, "hbm")
    t.wrap(ops, "interact_dot_backward", inter_bwd_bytes, "hbm")
    return t


KERNEL_OF = {
    "linear_forward": "%s, operands RC,RC (x W^T + b, ReLU)",
    "linear_backward_data": "%s, operands RC,RS (dz W, ReLU mask)",
    "linear_backward_weight": "%s, operands RS,RS, split-batch + colsum (dz^T x); partial sums of a stack's layers added by one "
                              "reduce_slabs_many launch (its time is included)",
    "embbag_forward": "embbag_fwd_group_kernel",
    "embbag_backward_update": "sparse_update_kernel + combine_runs_kernel (segmented reduction + fused optimizer)",
    "embbag_backward_prepare": "expand_pairs + the library's radix sort, digit_hist / digit_offsets / digit_scatter (ids only; "
                               "overlapped with the forward pass)",
    "interact_dot_forward": "interact_dot_fwd_kernel (interact_dot_fwd_gather_kernel: reads the table rows itself when every "
                            "bag holds one id)",
    "interact_dot_backward": "interact_dot_bwd_kernel (interact_dot_bwd_gather_kernel: gathers the table rows again when the forward kernel did)",
    "linear_forward_skinny": "skinny_k_fwd_kernel (13 input columns) / gemv_fwd_loss_bwd_kernel (one output column: prediction, loss gradient and input gradient)",
    "linear_backward_data_skinny": "gemv_bwd_data_kernel (one output column)",
    "linear_backward_weight_skinny": "skinny_k_bwd_weight_kernel / gemv_bwd_weight_kernel + reduce_slabs",
    "comm_forward_exchange": "RCCL all-to-all of pooled vectors (reduce-scatter for row-wise tables), peers' arrival included",
    "comm_backward_exchange": "RCCL all-to-all of their gradients (all-gather for row-wise tables)",
    "comm_all_reduce": "RCCL all-reduce of the dense gradients (with --two-comms: on a second communicator, beside the backward pass)",
}



def run_cpu_baseline(cfg):
    """The same operator sequence on libtorch's CPU kernels (oracle/cpu_baseline.py, kind "port"), timed on this
    box's host cores on a bounded sample of the workload: all cores, and one thread."""
    from oracle.cpu_baseline import time_cpu_baseline
    return time_cpu_baseline(cfg, batch=2048 if cfg.name == "criteo-kaggle" else 65536)


def layer_shapes(cfg):
    bsz = [cfg.dense_in] + list(cfg.bottom)
    tsz = [cfg.top_in] + list(cfg.top)
    return list(zip(bsz[:-1], bsz[1:])), list(zip(tsz[:-1], tsz[1:]))


def fused_last_layer(layers, i):
    """model.hip's rule (DLRM_FUSE_LAST=0 switches it off): the last of at least two layers, one output, <= 1024 inputs"""
    K, N = layers[i]
    return os.environ.get("DLRM_FUSE_LAST", "1")[:1] != "0" and i == len(layers) - 1 and i >= 1 and N == 1 and K <= 1024 and K % 4 == 0


def step_work(cfg, B, Bg, world, Tl, Trw, uniq, uniq_rw, idx_esz, row_esz, gather=False):
    """Algorithmic work of one step on one rank per operator class: (work, "mfma" flops | "hbm" bytes).
    gather: one id per bag, the interaction's forward kernel reads the table rows itself (no EmbeddingBag forward
    launch): it is priced with the rows, ids and offsets it reads and the copy of the block it leaves for the
    backward pass."""
    skinny = lambda K, N: (K <= 16 and N >= 64) or N == 1
    w = {k: 0.0 for k in ("linear_forward", "linear_backward_data", "linear_backward_weight", "linear_forward_skinny",
                          "linear_backward_data_skinny", "linear_backward_weight_skinny")}
    bot, top = layer_shapes(cfg)
    for stack, layers in (("bot", bot), ("top", top)):
        for i, (K, N) in enumerate(layers):
            # (a training step's last layer: prediction, loss gradient and input gradient in one launch that reads
            # the input, the labels and writes the three results)
            fused = stack == "top" and fused_last_layer(layers, i)
            if skinny(K, N):
                w["linear_forward_skinny"] += 4.0 * B * ((2 * K + 3) if fused else (N + K))
                w["linear_backward_weight_skinny"] += 4.0 * B * (N + K)
            else:
                w["linear_forward"] += 2.0 * B * N * K
                w["linear_backward_weight"] += 2.0 * B * N * K
            has_dgrad = i > 0 or stack == "top"      # the bottom MLP's first layer needs no input gradient
            if has_dgrad:
                masked = not (stack == "top" and i == 0)
                if fused:
                    pass
                elif N == 1:
                    w["linear_backward_data_skinny"] += 4.0 * B * (N + K * (2 if masked else 1))
                else:
                    w["linear_backward_data"] += 2.0 * B * N * K
    D, L = cfg.dim, cfg.lookups
    n_tw, n_rw = Bg * Tl * L, Bg * Trw * L
    out = {k: (v, "hbm" if k.endswith("_skinny") else "mfma") for k, v in w.items() if v}
    if not gather:
        out["embbag_forward"] = ((n_tw + n_rw / world) * D * row_esz + Bg * (Tl + Trw) * D * 4 + (n_tw + n_rw) * idx_esz +
                                 (Bg * (Tl + Trw) + 1) * idx_esz, "hbm")
    out["embbag_backward_update"] = (Bg * (Tl + Trw) * D * 4 + 2 * (uniq + uniq_rw) * D * row_esz + (n_tw + n_rw) * idx_esz, "hbm")
    out["embbag_backward_prepare"] = (0.0, "hbm")
    if cfg.interaction == "dot":
        F = cfg.num_features
        # gather: the forward kernel also reads a bag's id and offset; the backward kernel gathers the rows again and
        # reads them as well (DLRM_GATHER_BWD=0: the forward kernel leaves a copy of the [F, D] block for it instead)
        ids = 2 * (F - 1) * idx_esz if gather else 0
        gather_bwd = gather and os.environ.get("DLRM_GATHER_BWD", "1")[:1] != "0"
        copy = F * D * 4 if gather and not gather_bwd else 0
        out["interact_dot_forward"] = (B * ((F * D + D + F * (F - 1) // 2) * 4 + ids + copy), "hbm")
        out["interact_dot_backward"] = (B * ((2 * F * D + D + F * (F - 1) // 2) * 4 + (ids if gather_bwd else 0)), "hbm")
    return out


def kernel_rows(timing, work, steps, gemm_name, gemm_peak):
    """timing: {class: (ms summed, launches)} over `steps` steps -> the `kernels` rows of the bench line."""
    timing = dict(timing)
    red = timing.pop("linear_backward_weight_reduce", None)
    if red and "linear_backward_weight" in timing:
        # the stacks' reductions of the partial sums are part of the weight gradients' cost (their launches keep the count)
        ms, n = timing["linear_backward_weight"]
        timing["linear_backward_weight"] = (ms + red[0], n)
    rows = []
    for name, (ms, launches) in timing.items():
        wk, kind = work.get(name, (0.0, "hbm"))
        sec = ms * 1e-3
        ach = wk * steps / sec if sec > 0 else 0.0
        peak = gemm_peak if kind == "mfma" else HBM_PEAK
        kname = KERNEL_OF.get(name, name)
        if kind == "mfma":
            kname = kname % gemm_name
        rows.append({"op": name, "kernel": kname, "bound": kind, "achieved": ach / 1e12, "peak": peak / 1e12,
                     "unit": "TFLOP/s" if kind == "mfma" else "TB/s", "frac": ach / peak, "ms_per_step": ms / steps,
                     "launches_per_step": launches / steps, "avg_launch_ms": ms / max(launches, 1)})
    rows.sort(key=lambda k: -k["ms_per_step"])
    return rows


def roofline_of(dom, gemm, traffic):
    r = {"kernel": dom["kernel"], "bound": dom["bound"],
         "achieved": dom["achieved"] * (1e3 if dom["bound"] == "hbm" else 1.0),
         "peak": dom["peak"] * (1e3 if dom["bound"] == "hbm" else 1.0),
         "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s", "frac": dom["frac"],
         "avg_launch_ms": dom["avg_launch_ms"], "traffic": traffic.get(dom["op"]) if traffic else None}
    if dom["bound"] == "mfma":
        r["peak_note"] = GEMM_CHOICES[gemm][4]
        r["mfma_tflops_executed"] = dom["achieved"] * GEMM_CHOICES[gemm][3]
    return r


def load_traffic(world, workload):
    """HBM bytes per operator launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) of the
    same workload ("config:global batch") on one GPU; nothing for any other run."""
    for name in ("round3_hbm_traffic.json", "round2_hbm_traffic.json"):      # the newest pass of this workload
        tpath = os.path.join(ROOT, "profiles", name)
        if world == 1 and os.path.exists(tpath):
            d = json.load(open(tpath))
            if d.get("workload") == workload:
                return d.get("per_op_bytes_per_launch", {})
    return {}


def run_c_host(args, cfg, Bg, B, rank, world, dev):
    """The step hosted inside libdlrm_amd.so (dlrm_model_*); exchanges over dlrm_comm (RCCL)."""
    import torch.distributed as dist
    from dlrm_amd import comm as dcomm
    from dlrm_amd import ops
    from dlrm_amd.cmodel import CModel
    from dlrm_amd.data import synthetic_batch
    gemm = gemm_choice(args, B)
    gemm_name, gemm_peak = GEMM_CHOICES[gemm][1], GEMM_CHOICES[gemm][2]
    cm, cm2, owner = None, None, None
    if world == 1 and args.force_exchange:
        cm = dcomm.Comm(dcomm.unique_id(), 0, 1)
        cm2 = dcomm.Comm(dcomm.unique_id(), 0, 1) if args.two_comms else None
    if world > 1:
        # (--share-gpu: ranks share a device, which RCCL refuses: dlrm_comm's rehearsal transport carries the calls)
        # --two-comms: RCCL serialises the operations of one communicator; with a second one the all-reduce of the
        # dense gradients runs beside the backward exchange of the embedding side
        new_id = dcomm.unique_id_rehearsal if args.share_gpu else dcomm.unique_id
        ids = [(new_id(), new_id() if args.two_comms else None) if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        cm = dcomm.Comm(ids[0][0], rank, world)
        cm2 = None if ids[0][1] is None else dcomm.Comm(ids[0][1], rank, world)
        if args.sharding == "hybrid":
            from dlrm_amd.sharding import plan_hybrid
            owner = plan_hybrid(cfg.table_rows, world, cfg.dim, row_wise_min_rows=args.row_wise_min_rows).owner
    # one id per bag (Criteo's one-hot features): a single-GPU model reads the rows inside the interaction's forward kernel
    gather = (world == 1 and cfg.lookups == 1 and cfg.interaction == "dot" and cfg.row_dtype != "bf16" and
              not args.no_gather and not args.force_exchange)
    model = CModel(cfg, B, rank=rank, world=world, comm=cm, table_owner=owner, seed=0,
                   overlap=0 if args.no_overlap else (1 if args.overlap else -1), presort=0 if args.no_presort else 1,
                   wgrad_stream={"auto": -1, "on": 1, "off": 0}[args.wgrad_stream], device=dev,
                   gemm_mode=GEMM_CHOICES[gemm][0], one_id_per_bag=gather, force_exchange=args.force_exchange,
                   graph_steps=args.graph and world == 1, comm_dense=cm2)
    if args.graph:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))    # a graph cannot be captured on the NULL stream
    model.init_dense(0)
    Tl = model.info.num_whole_tables
    Trw = model.info.num_local_tables - Tl
    local_rows = [cfg.table_rows[g] for g, _, _, _ in model.local_tables]
    n_batches = args.batches or 4
    batches = []
    for i in range(n_batches):
        dense, idx, off, labels = synthetic_batch(local_rows, Bg, cfg.lookups, cfg.dense_in, dev, seed=1000 * (i + 1) + rank,
                                                  loss=cfg.loss, dist=args.dist, local_batch=B, shared_from=Tl,
                                                  shared_seed=77 * (i + 1))
        uniq, uniq_rw, n_per = 0, 0, Bg * cfg.lookups
        for lt, (g, _, rows, lo) in enumerate(model.local_tables):
            u = torch.unique(idx[lt * n_per:(lt + 1) * n_per])
            if lt < Tl:
                uniq += int(u.numel())
            else:
                uniq_rw += int(((u >= lo) & (u < lo + rows)).sum().item())
        batches.append((dense.contiguous(), idx, off, labels, (uniq, uniq_rw)))

    def step(i):
        dense, idx, off, labels, _ = batches[i % n_batches]
        model.load_batch(dense, labels)
        return model.train_step(idx, off)

    def barrier():
        if world > 1:
            dist.barrier()

    def work_for(step_ids):
        u = sum(batches[i % n_batches][4][0] for i in step_ids) / max(len(step_ids), 1)
        ur = sum(batches[i % n_batches][4][1] for i in step_ids) / max(len(step_ids), 1)
        return step_work(cfg, B, Bg, world, Tl, Trw, u, ur, 4, 2 if cfg.row_dtype == "bf16" else 4, gather)

    launches_per_step = 80
    # warm-up; its last steps time every class to find the one that dominates a step
    n_probe = min(2, args.warmup)
    for i in range(args.warmup - n_probe):
        step(i)
    model.enable_timing("all", launches_per_step * max(n_probe, 1))
    for i in range(args.warmup - n_probe, args.warmup):
        step(i)
    probe = model.collect_timing()
    timed_cls = [k for k, _ in sorted(probe.items(), key=lambda kv: -kv[1][0])
                 if k not in ("embbag_backward_prepare", "linear_backward_weight_reduce")][:1] or ["linear_backward_data"]
    if timed_cls == ["linear_backward_weight"]:
        timed_cls.append("linear_backward_weight_reduce")      # (part of that class's cost: kernel_rows adds it in)
    model.enable_timing([] if args.graph else timed_cls, 16 * args.steps)
    if args.graph:
        for i in range(n_batches):      # capture every resident batch before the timed region
            step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(model.err_flag.item()) != 0:
        raise SystemExit("out-of-range embedding index reported by the kernels")
    dt_rank = dt
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1e3 / args.steps
    value = Bg * args.steps / dt
    loss_value = float(loss.item())
    timed_ids = [args.warmup + i for i in range(args.steps)]
    dom_rows = kernel_rows(model.collect_timing(), work_for(timed_ids), args.steps, gemm_name, gemm_peak)
    traffic = load_traffic(world, "%s:%d" % (args.config, Bg))
    roofline = roofline_of(dom_rows[0], gemm, traffic) if dom_rows else None
    if roofline:
        roofline["timed"] = "HIP events around this operator's launches inside the timed steps"

    # every class, in extra steps after the timed region
    n_k = max(min(args.steps, 10), 1)
    model.enable_timing("all", launches_per_step * n_k)
    k_ids = [args.warmup + args.steps + i for i in range(n_k)]
    for i in k_ids:
        step(i)
    kernels = kernel_rows(model.collect_timing(), work_for(k_ids), n_k, gemm_name, gemm_peak)
    for k in kernels:
        k["traffic"] = traffic.get(k["op"]) if traffic else None
    model.enable_timing([], 0)
    # what every rank saw: its own time for the timed steps, its operator classes and collectives (events on the
    # streams they run on; a collective's time includes waiting for the slowest peer) and the bytes it exchanges
    per_rank = None
    if world > 1 or args.force_exchange:
        D4 = cfg.dim * 4
        mine = {"rank": rank, "ms_per_step": dt_rank * 1e3 / args.steps, "whole_tables": Tl, "row_wise_tables": Trw,
                "exchange_bytes_per_step": {
                    "forward_all_to_all_sent": Bg * Tl * D4, "backward_all_to_all_sent": Bg * Tl * D4,
                    "of_which_leave_the_gpu": Bg * Tl * D4 * (world - 1) // world,
                    "row_wise_reduce_scatter_in": Bg * Trw * D4, "row_wise_all_gather_out": Bg * Trw * D4,
                    "dense_all_reduce": int(model.info.n_params) * 4},
                "ms_per_step_by_class": {k["op"]: round(k["ms_per_step"], 4) for k in kernels}}
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
    if roofline is None and kernels:     # (--graph: nothing is bracketed inside a replayed graph)
        roofline = roofline_of(kernels[0], gemm, traffic)
        roofline["timed"] = "HIP events around this operator's launches in %d directly launched steps after the timed region" % n_k
    emb = next((k for k in kernels if k["op"] == "embbag_forward"), None)
    emb_note = "EmbeddingBag forward launches inside the step"
    if emb is None and gather:
        # no EmbeddingBag forward launch inside this step: the same batches through the kernel on its own
        tro = torch.tensor([f for _, f, _, _ in model.local_tables] + [model.info.total_rows], dtype=torch.int64, device=dev)
        scratch = torch.empty((Bg, Tl * cfg.dim), dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run = lambda i: ops.embbag_forward(model.weight, batches[i % n_batches][1], batches[i % n_batches][2], Bg, Tl, tro,
                                           include_last_offset=True, out=scratch, out_ld=Tl * cfg.dim)
        for i in range(2):
            run(i)
        e0.record()
        for i in range(10):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        wk = step_work(cfg, B, Bg, world, Tl, Trw, 0, 0, 4, 4, False)["embbag_forward"][0]
        emb = {"op": "embbag_forward", "kernel": KERNEL_OF["embbag_forward"], "bound": "hbm", "achieved": wk / (ms * 1e-3) / 1e12,
               "peak": HBM_PEAK / 1e12, "unit": "TB/s", "frac": wk / (ms * 1e-3) / HBM_PEAK, "ms_per_step": 0.0,
               "launches_per_step": 0.0, "avg_launch_ms": ms, "traffic": traffic.get("embbag_forward") if traffic else None,
               "note": "not part of this step (the interaction's forward kernel reads the rows): dlrm_embbag_forward alone "
                       "on the same batches, 10 launches"}
        kernels.append(emb)
        emb_note = emb["note"]
        del scratch

    # forward only (inference), same batches
    infer = None
    if not args.no_inference:
        n_inf = max(min(args.steps, 10), 1)
        for i in range(2):
            model.forward(batches[i % n_batches][1], batches[i % n_batches][2])
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for i in range(n_inf):
            dense, idx, off, labels, _ = batches[i % n_batches]
            model.load_batch(dense, labels)
            model.forward(idx, off)
        torch.cuda.synchronize()
        barrier()
        dt_inf = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_inf], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_inf = float(tt.item())
        infer = {"samples_per_s": Bg * n_inf / dt_inf, "ms_per_batch": dt_inf * 1e3 / n_inf, "batches": n_inf,
                 "launch": "dlrm_model_forward"}

    # The same steps with the layers' products on bf16 pieces -- the evaluation whose split of the operands is exact
    # (24 mantissa bits) -- measured in the same process right behind the run above, so that the line carries both.
    exact_ref = None
    if world == 1 and gemm == "f16x2" and not args.no_exact_reference and not args.graph:
        model.close()
        model = CModel(cfg, B, rank=rank, world=world, comm=cm, table_owner=owner, seed=0,
                       overlap=0 if args.no_overlap else (1 if args.overlap else -1), presort=0 if args.no_presort else 1,
                       wgrad_stream={"auto": -1, "on": 1, "off": 0}[args.wgrad_stream], device=dev,
                       gemm_mode=GEMM_CHOICES["bf16x3"][0], one_id_per_bag=gather, force_exchange=args.force_exchange)
        model.init_dense(0)
        for i in range(args.warmup):
            step(i)
        model.enable_timing(timed_cls, 16 * args.steps)      # the same class as in the main run, inside the timed steps
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(args.steps):
            ref_loss = step(args.warmup + i)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        ref_rows = kernel_rows(model.collect_timing(), work_for(timed_ids), args.steps, GEMM_CHOICES["bf16x3"][1],
                               GEMM_CHOICES["bf16x3"][2])
        ref_roofline = roofline_of(ref_rows[0], "bf16x3", {}) if ref_rows else None
        if ref_roofline:
            ref_roofline["timed"] = "HIP events around this operator's launches inside the timed steps"
        exact_ref = {"gemm": GEMM_CHOICES["bf16x3"][5], "value": Bg * args.steps / dt2, "unit": "samples/s",
                     "ms_per_step": dt2 * 1e3 / args.steps, "steps": args.steps, "warmup": args.warmup,
                     "loss": float(ref_loss.item()), "roofline": ref_roofline,
                     "note": "same model, batches, stream options and step count, no operator timing inside the steps.  The "
                             "two losses are not a parity check: with the reference's unscaled N(0,1) layer weights this "
                             "synthetic run is ill-conditioned (the loss of step 1 is in the hundreds) and runs that differ "
                             "in the last bit drift apart within a few steps -- the float32 MFMA evaluation ends at yet "
                             "another value; parity is what tests/ measure, per operator and per step against the oracle"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(cfg)
    if rank == 0:
        line = {
            "metric": "train samples/s (node) MLPerf-DLRM shape @1/2/4/8 GPU; EmbBag HBM GB/s",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (%s ids, random-init weights)" % args.dist,
            "config": {"workload": cfg.name, "global_batch": Bg, "tables": cfg.num_tables, "dim": cfg.dim,
                       "total_rows": sum(cfg.table_rows), "lookups_per_bag": cfg.lookups,
                       "bottom_mlp": [cfg.dense_in] + cfg.bottom, "top_mlp": [cfg.top_in] + cfg.top,
                       "interaction": cfg.interaction, "loss": cfg.loss, "emb_optimizer": cfg.emb_optimizer,
                       "gemm": GEMM_CHOICES[gemm][5],
                       "launch": "step hosted in libdlrm_amd.so (dlrm_model_train_step); exchanges over dlrm_comm (RCCL)" +
                       ("; captured into a HIP graph per resident batch and replayed" if args.graph and world == 1 else ""),
                       "embedding_lookup": "inside the interaction's forward kernel (one id per bag)" if gather else
                       "EmbeddingBag kernel into a [B, T, D] block",
                       "parallelism": ("single GPU, sharded schedule with one peer" if args.force_exchange else "single GPU")
                       if world == 1 else
                       ("%s x%d + data-parallel MLP" % ("table-wise" if not Trw else
                                                        "table-wise + %d row-wise tables" % Trw, world))},
            "loss": loss_value, "cache": None, "inference": infer, "roofline": roofline,
            "embbag_hbm_gbps": None if emb is None else emb["achieved"] * 1e3, "embbag_hbm_gbps_from": emb_note,
            "kernels": kernels, "kernels_timed": "every class, in %d extra steps after the timed region" % n_k,
            "cpu_baseline": cpu_baseline, "exact_split_reference": exact_ref, "ranks": per_rank,
        }
        print(json.dumps(line))
    model.close()
    if cm2 is not None:
        cm2.close()
    if cm is not None:
        cm.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=65536)
    ap.add_argument("--config", default="mlperf", choices=["mlperf", "kaggle"])
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--row-cap", type=int, default=0, help="cap rows per table (debug on small boxes); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-comms", action="store_true",
                    help="sharded C-hosted step: a second communicator for the all-reduce of the dense gradients, which then "
                         "starts behind the top MLP's backward pass and runs beside the backward exchange.  Not the default: "
                         "two communicators whose kernels wait for peers, on streams that can share a hardware queue (more "
                         "streams than queues), can end up queued in different orders on different ranks; with one "
                         "communicator RCCL keeps every rank's operations in issue order")
    ap.add_argument("--no-inference", action="store_true",
                    help="skip the forward-only batches after the train steps (profiler passes that average per launch)")
    ap.add_argument("--sharding", default="table", choices=["table", "hybrid"],
                    help="N>1: whole tables per rank, or the biggest tables split by rows over all ranks")
    ap.add_argument("--row-wise-min-rows", type=int, default=20_000_000)
    ap.add_argument("--cache-fraction", type=float, default=0.0,
                    help="tables in pinned host DRAM behind an HBM cache of this fraction of their pages (0 = tables in HBM)")
    ap.add_argument("--cache-rows-per-page", type=int, default=1)
    ap.add_argument("--cache-policy", default="lru", choices=["lru", "vec"],
                    help="lru: page / row LRU cache; vec: rows listed oldest request first, evicted rows repacked into new "
                         "pages (the reference's second policy)")
    ap.add_argument("--no-prefetch", action="store_true", help="cached mode: look ids up inside the step")
    ap.add_argument("--batches", type=int, default=0, help="distinct synthetic batches to cycle (default 4; 64 cached)")
    ap.add_argument("--host", default="auto", choices=["auto", "c", "python"],
                    help="who hosts the step: c = dlrm_model_* inside the library (default unless a cache is asked for), "
                         "python = dlrm_amd.model")
    ap.add_argument("--eager", action="store_true",
                    help="launch every operator from Python instead of replaying the recorded step (dlrm_plan)")
    ap.add_argument("--replay", action="store_true", help="(default) replay the recorded step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: collectives backend (gloo only to rehearse the sharded step where ranks share a GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: rank r uses device r %% device_count")
    ap.add_argument("--graph", action="store_true",
                    help="C host, N = 1: the step is captured into a HIP graph per resident batch and replayed "
                         "(dlrm_model_config.graph_steps); the operator classes are then timed only after the timed region")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1, C host: run the sharded schedule (every exchange, with one peer, over RCCL) to price what "
                         "the schedule itself costs beside the link time of a real multi-GPU run")
    ap.add_argument("--no-gather", action="store_true",
                    help="single GPU, one id per bag: pool the rows with the EmbeddingBag kernel into a [B, T, D] block "
                         "instead of reading them inside the interaction kernels")
    ap.add_argument("--no-overlap", action="store_true", help="embedding side on the caller's stream")
    ap.add_argument("--overlap", action="store_true", help="embedding side on its own stream (default for N>1)")
    ap.add_argument("--no-presort", action="store_true", help="sort the occurrences inside the sparse update")
    ap.add_argument("--gemm", default="auto", choices=["auto", "f32", "bf16x3", "f16x2"],
                    help="evaluation of the big layers' float32 products (auto: f16x2 from a local batch of 8192 up, else bf16x3)")
    ap.add_argument("--lr", type=float, default=0.0,
                    help="learning rate (default: the configuration's, 0.01 like the reference's SGD; on random labels "
                         "that run is chaotic and the loss after a few steps says nothing, a small rate keeps it smooth)")
    ap.add_argument("--no-exact-reference", action="store_true",
                    help="skip the second, shorter measurement of the same steps on bf16 pieces (f16x2 runs only)")
    ap.add_argument("--wgrad-stream", default="auto", choices=["auto", "on", "off"],
                    help="MLP weight gradients on their own stream beside the data gradients (auto: local batch <= 32768)")
    ap.add_argument("--progress", action="store_true", help="synchronise and print a line to stderr after every step")
    ap.add_argument("--watchdog", type=int, default=0,
                    help="seconds after which a run that has not finished dumps every thread's stack and exits")
    args = ap.parse_args()
    if args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(900, exit=True)     # a stuck collective ends with every thread's stack
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from dlrm_amd import ops
    from dlrm_amd.data import synthetic_batch
    from dlrm_amd.model import DLRM, DLRMConfig

    cfg = DLRMConfig.mlperf() if args.config == "mlperf" else DLRMConfig.kaggle()
    if args.lr > 0:
        cfg.lr = args.lr
    if args.row_cap:
        cfg.table_rows = [min(r, args.row_cap) for r in cfg.table_rows]
    Bg = args.global_batch
    if args.config == "kaggle" and args.global_batch == 65536:
        Bg = 2048
    assert Bg % world == 0
    B = Bg // world
    host = args.host
    if host == "auto":
        host = "python" if (args.cache_fraction > 0 or args.eager) else "c"
    if host == "c":
        if args.cache_fraction > 0:
            raise SystemExit("the cached modes are hosted by dlrm_amd.model: --host python")
        return run_c_host(args, cfg, Bg, B, rank, world, dev)
    model_ref = {"unique_rows": 0, "unique_rows_rw": 0, "lookups": cfg.lookups, "world": world}
    timer = install_timers(ops, model_ref)
    gemm = gemm_choice(args, B)
    if gemm == "f16x2" and args.gemm == "auto" and ops.get_gemm_mode() != 2:
        gemm = "bf16x3"     # (the Python-hosted operators would measure every operand in a pass of its own)
    ops.set_gemm_mode({"f32": 0, "bf16x3": 1, "f16x2": 2}[gemm])
    gemm_name, gemm_peak = GEMM_CHOICES[gemm][1], GEMM_CHOICES[gemm][2]
    plan = None
    if args.sharding == "hybrid" and world > 1:
        from dlrm_amd.sharding import plan_hybrid
        plan = plan_hybrid(cfg.table_rows, world, cfg.dim, row_wise_min_rows=args.row_wise_min_rows)
    cached = args.cache_fraction > 0
    model = DLRM(cfg, B, device=dev, rank=rank, world=world, seed=0, process_group=pg, plan=plan,
                 cache_fraction=args.cache_fraction if cached else None,
                 cache_rows_per_page=args.cache_rows_per_page, replay=not args.eager, cache_policy=args.cache_policy,
                 cache_log_pages=((args.steps + args.warmup + 2) * Bg * cfg.lookups) // 8 + 64,
                 overlap=False if args.no_overlap else (True if args.overlap else None), presort=False if args.no_presort else None,
                 wgrad_stream={"auto": None, "on": True, "off": False}[args.wgrad_stream])
    local_rows = [cfg.table_rows[t] for t in model.local_table_ids]   # ids are global: full table sizes

    # a few different batches, resident in HBM, cycled
    n_batches = args.batches or (64 if cached else 4)
    batches = []
    for i in range(n_batches):
        dense, idx, off, labels = synthetic_batch(local_rows, Bg, cfg.lookups, cfg.dense_in, dev,
                                                  seed=1000 * (i + 1) + rank, loss=cfg.loss, dist=args.dist,
                                                  local_batch=B, shared_from=model.Tl, shared_seed=77 * (i + 1))
        # unique rows touched per batch (for the algorithmic bytes of the sparse update)
        uniq, uniq_rw = 0, 0
        n_per = Bg * cfg.lookups
        for lt, t in enumerate(model.local_table_ids):
            u = torch.unique(idx[lt * n_per:(lt + 1) * n_per])
            if lt < model.Tl:
                uniq += int(u.numel())
            else:
                lo, hi = model.plan.row_range(t, rank)
                uniq_rw += int(((u >= lo) & (u < hi)).sum().item())
        batches.append((dense, idx, off, labels, (uniq, uniq_rw)))

    def step(i):
        dense, idx, off, labels, (uniq, uniq_rw) = batches[i % n_batches]
        model_ref["unique_rows"], model_ref["unique_rows_rw"] = uniq, uniq_rw
        model.load_batch(dense, labels)
        if cached and not args.no_prefetch and args.cache_policy == "lru":
            out = model.train_step(idx, off, next_indices=batches[(i + 1) % n_batches][1])
        else:
            out = model.train_step(idx, off)
        if args.progress:
            torch.cuda.synchronize()
            print("[rank %d] step %d done at %.2f s" % (rank, i, time.perf_counter() - t_start), file=sys.stderr, flush=True)
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    t_start = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if cached:
        model.cache.counters(reset=True)
        cache_c0 = model.cache.counters()
    barrier()
    torch.cuda.synchronize()
    timer.enabled = args.eager
    if not args.eager:
        model._plan.enable_timing(args.steps)      # event pairs around every recorded entry, on its stream
    uniq_sum = [0, 0]
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        uniq_sum[0] += model_ref["unique_rows"]
        uniq_sum[1] += model_ref["unique_rows_rw"]
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if int(model.err_flag.item()) != 0:
        raise SystemExit("out-of-range embedding index reported by the kernels")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1e3 / args.steps
    value = Bg * args.steps / dt

    # per-kernel-class accounting from the events recorded inside the timed steps
    summ = timer.summary()
    if not args.eager:
        # the recorded step timed itself: milliseconds per entry summed over the timed replays
        model_ref["unique_rows"] = uniq_sum[0] / args.steps
        model_ref["unique_rows_rw"] = uniq_sum[1] / args.steps
        ms, runs = model._plan.collect_timing()
        for i, (name, a, kw) in enumerate(model._plan.meta):
            if name not in timer.work or runs[i] == 0:
                continue
            fn, kind = timer.work[name]
            # the second half of the sparse update carries the operator's algorithmic bytes; its first
            # half (the sort, beside the forward pass on its own stream) is listed on its own: sort
            # scratch traffic is overhead, not algorithmic work
            key = "embbag_backward_update" if name == "embbag_backward_apply" else name
            key, work, kind = classify(key, a, kw, fn(*a, **kw), kind)
            d = summ.setdefault(key, {"ms": 0.0, "work": 0.0, "launches": 0, "kind": kind})
            d["ms"] += ms[i]
            d["work"] += work * runs[i]
            d["launches"] += runs[i]
    kernels = []
    for name, d in summ.items():
        sec = d["ms"] * 1e-3
        ach = d["work"] / sec if sec > 0 else 0.0
        peak = gemm_peak if d["kind"] == "mfma" else HBM_PEAK
        kname = KERNEL_OF.get(name, name)
        if d["kind"] == "mfma":
            kname = kname % gemm_name
        kernels.append({
            "op": name, "kernel": kname, "bound": d["kind"],
            "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s" if d["kind"] == "mfma" else "TB/s",
            "frac": ach / peak, "ms_per_step": d["ms"] / args.steps, "launches_per_step": d["launches"] / args.steps,
            "avg_launch_ms": d["ms"] / max(d["launches"], 1),
        })
    kernels.sort(key=lambda k: -k["ms_per_step"])
    dom = kernels[0] if kernels else None
    roofline = None
    if dom:
        roofline = {"kernel": dom["kernel"], "bound": dom["bound"],
                    "achieved": dom["achieved"] * (1e3 if dom["bound"] == "hbm" else 1.0),
                    "peak": dom["peak"] * (1e3 if dom["bound"] == "hbm" else 1.0),
                    "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s", "frac": dom["frac"],
                    "avg_launch_ms": dom["avg_launch_ms"], "traffic": None}
        if dom["bound"] == "mfma":
            roofline["peak_note"] = GEMM_CHOICES[gemm][4]
            roofline["mfma_tflops_executed"] = dom["achieved"] * GEMM_CHOICES[gemm][3]
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
        traffic = {} if cached else load_traffic(world, "%s:%d" % (args.config, Bg))
        if traffic:
            for k in kernels:
                k["traffic"] = traffic.get(k["op"])
            roofline["traffic"] = traffic.get(dom["op"])
    emb = next((k for k in kernels if k["op"] == "embbag_forward"), None)

    cache_info = None
    if cached:
        c1, st = model.cache.counters(), model.cache.state()
        hit = int((c1["hit"] - cache_c0["hit"]).sum())
        miss = int((c1["miss"] - cache_c0["miss"]).sum())
        vec = args.cache_policy == "vec"
        page_bytes = 4096 if vec else args.cache_rows_per_page * cfg.dim * 4
        cache_info = {"policy": args.cache_policy, "fraction": args.cache_fraction,
                      "rows_per_page": 4096 // (cfg.dim * 4) if vec else args.cache_rows_per_page,
                      "prefetch": not args.no_prefetch and not vec, "hit_rate_lookups": hit / max(hit + miss, 1),
                      "pages_read_per_step": int(c1["read"].sum()) / args.steps,
                      "pages_written_back_per_step": int(c1["write"].sum()) / args.steps,
                      "h2d_GBps": int(c1["read"].sum()) * page_bytes / dt / 1e9,
                      "cache_GB": (model.cache.num_slots * cfg.dim * 4 if vec else st.cache_size * page_bytes) / 1e9,
                      "table_GB": sum(cfg.table_rows[t] for t in model.local_table_ids) * cfg.dim * 4 / 1e9,
                      "error_flags": int(st.error_flags)}

    # forward only (inference), same batches, after the timed training steps
    infer = None
    if not cached:
        n_inf = max(min(args.steps, 10), 1)
        for i in range(2):
            model.forward(batches[i % n_batches][1], batches[i % n_batches][2])
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for i in range(n_inf):
            dense, idx, off, labels, _ = batches[i % n_batches]
            model.load_batch(dense, labels)
            model.forward(idx, off)
        torch.cuda.synchronize()
        barrier()
        dt_inf = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_inf], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt_inf = float(tt.item())
        infer = {"samples_per_s": Bg * n_inf / dt_inf, "ms_per_batch": dt_inf * 1e3 / n_inf, "batches": n_inf,
                 "launch": "eager (one Python call per operator)"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(cfg)

    if rank == 0:
        line = {
            "metric": "train samples/s (node) MLPerf-DLRM shape @1/2/4/8 GPU; EmbBag HBM GB/s",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (%s ids, random-init weights)" % args.dist,
            "config": {"workload": cfg.name, "global_batch": Bg, "tables": cfg.num_tables, "dim": cfg.dim,
                       "total_rows": sum(cfg.table_rows), "lookups_per_bag": cfg.lookups,
                       "bottom_mlp": [cfg.dense_in] + cfg.bottom, "top_mlp": [cfg.top_in] + cfg.top,
                       "interaction": cfg.interaction, "loss": cfg.loss, "emb_optimizer": cfg.emb_optimizer,
                       "gemm": GEMM_CHOICES[gemm][5],
                       "launch": "eager (one Python call per operator)" if args.eager else
                       "recorded step replayed through dlrm_plan_run",
                       "parallelism": "single GPU" if world == 1 else
                       ("%s x%d + data-parallel MLP" % ("table-wise" if not model.Trw else
                                                        "table-wise + %d row-wise tables" % model.Trw, world))},
            "loss": float(loss.item()),
            "cache": cache_info,
            "inference": infer,
            "roofline": roofline,
            "embbag_hbm_gbps": None if emb is None else emb["achieved"] * 1e3,
            "kernels": kernels,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
