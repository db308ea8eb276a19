"""world_size 2 on CPU (gloo): the sharded train step of dlrm_amd.model -- table-wise plan
(all-to-all of pooled vectors and of their gradients) and hybrid plan (plus row-wise split
tables: reduce-scatter of partial sums, all-gather of gradients), all-reduce of the MLP
gradients, global-mean loss -- must give what one process gives on the whole batch.  Compute is the CPU
oracle patched in for dlrm_amd.ops (tests/cpu_ops_shim.py): this exercises the exchange layout
and the reduction logic, which is the part no single-GPU test can see."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dlrm_amd.sharding import plan_hybrid, plan_table_wise


def test_plan_table_wise_balances_and_is_deterministic():
    from dlrm_amd.model import MLPERF_TABLE_ROWS
    p = plan_table_wise(MLPERF_TABLE_ROWS, 8, dim=128)
    assert sorted(t for r in p.tables_of_rank for t in r) == list(range(26))
    assert p == plan_table_wise(MLPERF_TABLE_ROWS, 8, dim=128)
    counts = [len(r) for r in p.tables_of_rank]
    assert max(counts) - min(counts) <= 1                     # equal lookups per table: traffic is balanced
    mem = [sum(MLPERF_TABLE_ROWS[t] for t in r) for r in p.tables_of_rank]
    assert max(mem) < 2 * 40_000_000                          # no rank holds two of the capped tables
    with pytest.raises(ValueError):
        plan_table_wise(MLPERF_TABLE_ROWS, 8, dim=128, mem_cap_bytes=1e9)
    assert p.feature_order == [t for r in p.tables_of_rank for t in r]
    first = 0
    for r in range(8):
        assert p.feature_slot(r) == (first, counts[r])
        first += counts[r]
    one = plan_table_wise([5, 7, 9], 1)
    assert one.tables_of_rank == [[0, 1, 2]]
    many = plan_table_wise([5, 7], 4)            # more ranks than tables: some ranks own nothing
    assert sorted(len(r) for r in many.tables_of_rank) == [0, 0, 1, 1]


def test_plan_hybrid_splits_big_tables_by_rows():
    from dlrm_amd.model import MLPERF_TABLE_ROWS
    p = plan_hybrid(MLPERF_TABLE_ROWS, 8, dim=128, row_wise_min_rows=20_000_000)
    big = [t for t, r in enumerate(MLPERF_TABLE_ROWS) if r >= 20_000_000]
    assert p.row_wise == big and len(big) == 5
    for t in big:
        rr = p.row_ranges[t]
        assert rr[0][0] == 0 and rr[-1][1] == MLPERF_TABLE_ROWS[t]
        assert all(rr[i][1] == rr[i + 1][0] for i in range(7))
        assert max(hi - lo for lo, hi in rr) - min(hi - lo for lo, hi in rr) <= 1
        assert p.owner[t] == -1
    assert sorted(p.feature_order) == list(range(26)) and p.feature_order[-5:] == big
    assert p.n_table_wise == 21
    assert sorted(t for r in p.tables_of_rank for t in r) == [t for t in range(26) if t not in big]
    with pytest.raises(ValueError):
        plan_hybrid([3, 100], 4, row_wise_tables=[0])       # fewer rows than ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg(kind):
    from dlrm_amd.model import DLRMConfig
    if kind == "dot":
        return DLRMConfig(table_rows=[50, 7, 300, 20, 64], dim=16, dense_in=13, bottom=[32, 16], top=[24, 1], lr=0.05,
                          lookups=2)
    return DLRMConfig(table_rows=[40, 9, 100], dim=8, dense_in=6, bottom=[16, 8], top=[12, 1], interaction="cat",
                      loss="mse", final_relu=True, pooling="mean", lookups=3, lr=0.02)


def _worker(rank, world, port, kind, ret, device="cpu", replay=False, backend="gloo", force_exchange=False):
    """device "cpu": the CPU oracle stands in for dlrm_amd.ops.  device "cuda": the ranks share GPU 0 and run
    the HIP kernels; gloo moves the buffers through host memory (RCCL refuses two ranks on one device).
    backend "nccl" (RCCL) with world 1 and force_exchange: the sharded schedule with every collective on
    device buffers in stream order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dlrm_amd import ops
        if device == "cpu":
            from tests import cpu_ops_shim
            cpu_ops_shim.install(ops)
        else:
            torch.cuda.set_device(0)
        from dlrm_amd.model import DLRM
        from oracle import oracle as orc
        from tests.oracle_model import make_batch, oracle_train_step

        hybrid = kind.endswith("-rw")
        kind = kind.split("-")[0]
        cfg = _cfg(kind)
        Bl = 6
        Bg = Bl * world
        plan = None
        if hybrid:   # the two biggest tables are split by rows, the rest placed whole
            plan = plan_hybrid(cfg.table_rows, world, cfg.dim, row_wise_min_rows=60 if kind == "dot" else 40)
            assert len(plan.row_wise) == 2
        m = DLRM(cfg, Bl, device=device, rank=rank, world=world, seed=4, plan=plan, replay=replay,
                 force_exchange=force_exchange)
        if force_exchange:
            assert m.sharded and (device == "cpu" or m.side is not None)   # exchanges on the embedding side's own stream
        dev = m.dev
        host = lambda t: t.detach().cpu().numpy()
        if kind == "cat":
            for b in m.bottom.b + m.top.b:
                b.abs_()
        rng = np.random.default_rng(123)                      # every rank draws the same global batches
        # single-process reference state: all tables (global order), same dense parameters
        tables = [orc.table_init_uniform(r * cfg.dim, (4 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
                  for t, r in enumerate(cfg.table_rows)]
        Ws = [host(w).copy() for w in m.bottom.W + m.top.W]
        bs = [host(b).copy() for b in m.bottom.b + m.top.b]
        bad = []

        def close(name, a, r, tol=1e-5):
            """the single-GPU tests' metric: largest error relative to the tensor's largest reference value"""
            a, r = np.asarray(a, dtype=np.float64), np.asarray(r, dtype=np.float64)
            if np.abs(a - r).max() > tol * max(np.abs(r).max(), 1e-30):
                bad.append("%s step %d: max err %.3g (max |ref| %.3g)" % (name, step, np.abs(a - r).max(), np.abs(r).max()))
        for step in range(3 if replay else 2):           # replay: record, replay, replay with new ids
            dense, idx, off, labels = make_batch(rng, cfg, Bg)
            L = cfg.lookups
            # this rank's tables over the global batch
            my = m.local_table_ids
            idx_l = np.concatenate([idx[t * Bg * L:(t + 1) * Bg * L] for t in my]) if my else np.zeros(0, np.int64)
            off_l = np.arange(len(my) * Bg + 1, dtype=np.int64) * L
            sl = slice(rank * Bl, (rank + 1) * Bl)
            # the index exchange: from this rank's slice of the batch (all tables, plan order) to the ids of
            # its tables for the global batch -- must equal the direct construction above, bit for bit
            loc = np.concatenate([idx[t * Bg * L + rank * Bl * L:t * Bg * L + (rank + 1) * Bl * L]
                                  for t in m.plan.feature_order])
            x_idx, x_off = m.exchange_ids(torch.from_numpy(loc).to(dev))
            if not (np.array_equal(host(x_idx), idx_l) and np.array_equal(host(x_off), off_l)):
                bad.append("index exchange step %d" % step)
            m.load_batch(torch.from_numpy(dense[sl]).to(dev), torch.from_numpy(labels[sl]).to(dev))
            loss = m.train_step(torch.from_numpy(idx_l).to(dev), torch.from_numpy(off_l).to(dev))
            tot = loss.detach().clone() if backend == "nccl" else loss.detach().cpu().clone()
            dist.all_reduce(tot)                               # per-rank shares add up to the global mean
            loss_ref, pred_ref = oracle_train_step(orc, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                                   cfg.lr, feature_order=m.plan.feature_order)
            close("loss", np.float64(tot.item()), np.float64(loss_ref))
            close("pred", host(m.pred).reshape(-1), pred_ref[sl])
            tro = host(m.table_row_offsets)
            for lt, t in enumerate(my):
                lo, hi = (0, cfg.table_rows[t]) if lt < m.Tl else m.plan.row_range(t, rank)
                close("table %d" % t, host(m.weight[tro[lt]:tro[lt + 1]]), tables[t][lo:hi])
            for n, (a, r) in enumerate(zip(m.bottom.W + m.top.W + m.bottom.b + m.top.b, Ws + bs)):
                close("dense parameter %d" % n, host(a), r)
        ret[rank] = "; ".join(bad) if bad else True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["dot", "cat", "dot-rw", "cat-rw"])
def test_sharded_step_equals_single_process(kind):
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, kind, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("kind", ["dot-rw", "cat"])
def test_forced_exchange_with_one_rank(kind):
    """DLRM(force_exchange=True) on one rank runs the sharded schedule (every pack / exchange / unpack with
    one peer) and must still equal the single-process step."""
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(1, port, kind, ret, "cpu", False, "gloo", True), nprocs=1, join=True)
    assert dict(ret) == {0: True}
