"""The whole-model handle of the C ABI (dlrm_model_*: placement, streams, exchanges and operator order inside
libdlrm_amd.so, no Python in the step) against the same step assembled from the CPU oracle -- the comparison of
tests/test_model_gpu.py, driven through the entry points a C++ host calls.  Single GPU, and the sharded schedule
over a dlrm_comm (RCCL) communicator of one rank with whole and row-wise split tables.
Tolerance: 1e-5 relative to each tensor's largest |reference| value; loss 1e-5 relative."""
import numpy as np
import pytest
import torch

from tests.oracle_model import make_batch, oracle_train_step
from tests.test_embbag_forward_gpu import rel_err
from tests.test_model_gpu import small_cfgs

pytestmark = pytest.mark.gpu


def tables_of(m):
    out = {}
    for g, first, rows, lo in m.local_tables:
        out[g] = (m.weight[first:first + rows].float().cpu().numpy().copy(), lo)
    return out


def local_batch_ids(m, cfg, idx, off, B):
    """the ids of the model's local tables, in its order, out of a batch in global table order"""
    segs, lens = [], []
    for g, _, _, _ in m.local_tables:
        s, e = off[g * B], off[(g + 1) * B]
        segs.append(idx[s:e])
        lens.append(np.diff(off[g * B:(g + 1) * B + 1]))
    li = np.concatenate(segs) if segs else np.zeros(0, np.int64)
    lo = np.concatenate([[0], np.cumsum(np.concatenate(lens))]) if lens else np.zeros(1, np.int64)
    return li, lo


@pytest.mark.parametrize("name", ["mlperf_like", "kaggle_like", "rmc1_like", "adagrad_bf16rows_off"])
@pytest.mark.parametrize("mode", ["single", "exchange", "exchange-rw"])
@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("gemm", ["default", "f16x2"])
def test_c_hosted_train_step_matches_oracle(oracle, name, mode, ragged, gemm):
    """gemm = f16x2: the big layers' products on two f16 pieces per operand with the operands' maxima travelling from
    kernel to kernel (what the model picks by itself from a local batch of 8192 up), same tolerance."""
    from dlrm_amd import _capi, comm
    from dlrm_amd.cmodel import CModel
    if gemm == "f16x2" and (mode == "exchange-rw" or ragged):
        pytest.skip("the evaluation mode does not touch the sparse path: covered at the other corners")
    cfg = small_cfgs()[name]
    B, T = 37, cfg.num_tables
    cm, cm2, owner = None, None, None
    if mode != "single":
        cm = comm.Comm(comm.unique_id(), 0, 1)
        if ragged or gemm == "f16x2":  # ... with a second communicator for the dense gradients' all-reduce, which then
            cm2 = comm.Comm(comm.unique_id(), 0, 1)     # starts behind the top MLP's backward pass on a stream of its own
        if mode == "exchange-rw":      # the two largest tables are split by rows (one range: the only rank's)
            big = sorted(range(T), key=lambda t: -cfg.table_rows[t])[:2]
            owner = [-1 if t in big else 0 for t in range(T)]
    m = CModel(cfg, B, comm=cm, table_owner=owner, seed=3, force_exchange=mode != "single", idx_dtype=torch.int64,
               gemm_mode=_capi.GEMM_F16X2 if gemm == "f16x2" else 0, comm_dense=cm2)
    m.init_dense(7)
    Ws, bs = m.get_dense()
    if name == "rmc1_like":            # positive biases keep the all-ReLU reference model alive
        bs = [np.abs(b) for b in bs]
        m.set_dense(Ws, bs)
    got = tables_of(m)
    tables = [got[t][0] for t in range(T)]
    state = None if m.opt_state is None else np.zeros(sum(cfg.table_rows), np.float32)
    n_bottom = len(cfg.bottom)
    rng = np.random.default_rng(17)
    for step in range(3):
        dense, idx, off, labels = make_batch(rng, cfg, B, ragged)
        li, lo = local_batch_ids(m, cfg, idx, off, B)
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        loss = m.train_step(torch.from_numpy(li).cuda(), torch.from_numpy(lo).cuda())
        torch.cuda.synchronize()
        assert int(m.err_flag.item()) == 0
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, n_bottom, cfg.lr,
                                               feature_order=m.feature_order, emb_opt_state=state)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        for t, (a, _) in tables_of(m).items():
            assert rel_err(a, tables[t]) <= 1e-5, t
        gW, gb = m.get_dense()
        for a, r in zip(gW + gb, Ws + bs):
            assert rel_err(a, r) <= 1e-5
    # inference changes nothing and predicts what the oracle's forward pass predicts
    import copy
    dense, idx, off, labels = make_batch(rng, cfg, B, ragged)
    li, lo = local_batch_ids(m, cfg, idx, off, B)
    m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
    pred = m.forward(torch.from_numpy(li).cuda(), torch.from_numpy(lo).cuda())
    torch.cuda.synchronize()
    _, pred_ref = oracle_train_step(oracle, cfg, copy.deepcopy(tables), copy.deepcopy(Ws), copy.deepcopy(bs), dense, idx,
                                    off, labels, n_bottom, cfg.lr, feature_order=m.feature_order,
                                    emb_opt_state=None if state is None else state.copy())
    assert rel_err(pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
    m.close()
    if cm2 is not None:
        cm2.close()
    if cm is not None:
        cm.close()


@pytest.mark.parametrize("shape", ["mlperf_like", "kaggle", "mlperf_wide"])
def test_c_hosted_one_id_per_bag_reads_rows_in_the_interaction(oracle, shape):
    """dlrm_model_config.one_id_per_bag: the interaction's forward kernel reads the table rows itself instead of a
    pooled [B, T, dim] block (and leaves a copy of the block for the backward pass of a training step).  Same arithmetic on the same values, so predictions, loss, updated rows and dense parameters
    equal those of the pooling model BIT FOR BIT over three steps (and inference likewise); the first step's
    predictions also match the oracle to 1e-5.  Shapes: a small one, the exact Criteo-Kaggle model (26 tables,
    dim 16, batch 2048) and 26 tables of dim 128 (the MLPerf feature block) at a ragged batch."""
    from dlrm_amd.cmodel import CModel
    from dlrm_amd.model import DLRMConfig
    if shape == "mlperf_like":
        cfg, B = small_cfgs()["mlperf_like"], 37
    elif shape == "kaggle":
        from tests.test_config_shapes_gpu import kaggle_cfg
        cfg, B = kaggle_cfg(), 2048
    else:
        cfg, B = DLRMConfig(table_rows=[3, 50, 1000, 7] * 6 + [64, 20000], dim=128, dense_in=13, bottom=[64, 128],
                            top=[64, 1], lr=0.001), 333
    assert cfg.lookups == 1
    T = cfg.num_tables
    a = CModel(cfg, B, seed=3, one_id_per_bag=False)
    b = CModel(cfg, B, seed=3, one_id_per_bag=True)
    a.init_dense(7)
    b.init_dense(7)
    Ws, bs = a.get_dense()
    tables = [a.weight[f:f + r].float().cpu().numpy().copy() for _, f, r, _ in a.local_tables]
    rng = np.random.default_rng(23)
    for step in range(3):
        dense, idx, off, labels = make_batch(rng, cfg, B, False)
        d_idx, d_off = torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda()
        losses = []
        for m in (a, b):
            m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
            losses.append(m.train_step(d_idx, d_off).clone())
        torch.cuda.synchronize()
        assert int(a.err_flag.item()) == 0 and int(b.err_flag.item()) == 0
        assert torch.equal(losses[0], losses[1])
        assert torch.equal(a.pred, b.pred)
        assert torch.equal(a.weight, b.weight)
        assert torch.equal(a.params, b.params)
        if step == 0:
            _, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom), cfg.lr,
                                            feature_order=b.feature_order)
            assert rel_err(b.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
    dense, idx, off, labels = make_batch(rng, cfg, B, False)
    d_idx, d_off = torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(off.astype(np.int32)).cuda()
    preds = []
    for m in (a, b):
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        preds.append(m.forward(d_idx, d_off).clone())
    torch.cuda.synchronize()
    assert torch.equal(preds[0], preds[1])
    # a batch that breaks the promise is reported, not silently mis-pooled
    d_off2 = d_off.clone()
    d_off2[5] -= 1
    b.forward(d_idx, d_off2)
    torch.cuda.synchronize()
    assert int(b.err_flag.item()) == 2
    a.close()
    b.close()


def _rank_worker(rank, world, comm_id, comm_id2, name, rw, ret):
    """One rank of a C-hosted sharded run on GPU 0 (rehearsal transport of dlrm_comm: RCCL refuses ranks that share
    a device): its slice of the global batch, the ids of its tables for the whole batch, compared with the oracle's
    single-process step on the global batch."""
    try:
        from oracle import oracle
        from dlrm_amd import comm
        from dlrm_amd.cmodel import CModel
        torch.cuda.set_device(0)
        cfg = small_cfgs()[name]
        Bl, T = 12, cfg.num_tables
        Bg = Bl * world
        cm = comm.Comm(comm_id, rank, world)
        cm2 = comm.Comm(comm_id2, rank, world) if comm_id2 is not None else None
        owner = None
        if rw:                       # the two largest tables are split by rows over all ranks, the rest placed whole
            big = sorted(range(T), key=lambda t: -cfg.table_rows[t])[:2]
            rest = [t for t in range(T) if t not in big]
            owner = [-1 if t in big else rest.index(t) % world for t in range(T)]
        m = CModel(cfg, Bl, rank=rank, world=world, comm=cm, table_owner=owner, seed=3, idx_dtype=torch.int64, comm_dense=cm2)
        m.init_dense(7)
        Ws, bs = m.get_dense()
        if name == "rmc1_like":
            bs = [np.abs(b) for b in bs]
            m.set_dense(Ws, bs)
        tables = [oracle.table_init_uniform(r * cfg.dim, (3 << 20) + 7919 * (t + 1)).reshape(r, cfg.dim)
                  for t, r in enumerate(cfg.table_rows)]
        for t, (a, lo) in tables_of(m).items():          # every shard starts as the matching rows of the whole table
            assert np.array_equal(a, tables[t][lo:lo + a.shape[0]]), "initial rows of table %d" % t
        rng = np.random.default_rng(17)                  # every rank draws the same global batches
        bad = []
        for step in range(3):
            dense, idx, off, labels = make_batch(rng, cfg, Bg, True)
            li, lo = local_batch_ids(m, cfg, idx, off, Bg)
            sl = slice(rank * Bl, (rank + 1) * Bl)
            m.load_batch(torch.from_numpy(dense[sl].copy()).cuda(), torch.from_numpy(labels[sl].copy()).cuda())
            loss = m.train_step(torch.from_numpy(li).cuda(), torch.from_numpy(lo).cuda())
            torch.cuda.synchronize()
            if int(m.err_flag.item()) != 0:
                bad.append("error flag at step %d" % step)
            tot = loss.detach().clone()
            cm.all_reduce_sum_(tot)                      # per-rank shares add up to the global mean
            loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, len(cfg.bottom),
                                                   cfg.lr, feature_order=m.feature_order)
            if abs(tot.item() - loss_ref) > 1e-5 * abs(loss_ref) + 1e-7:
                bad.append("loss step %d: %r vs %r" % (step, tot.item(), loss_ref))
            if rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref[sl]) > 1e-5:
                bad.append("pred step %d" % step)
            for t, (a, lo_row) in tables_of(m).items():
                if rel_err(a, tables[t][lo_row:lo_row + a.shape[0]]) > 1e-5:
                    bad.append("table %d step %d" % (t, step))
            gW, gb = m.get_dense()
            for n, (a, r) in enumerate(zip(gW + gb, Ws + bs)):
                if rel_err(a, r) > 1e-5:
                    bad.append("dense parameter %d step %d" % (n, step))
        m.close()
        if cm2 is not None:
            cm2.close()
        cm.close()
        ret[rank] = "; ".join(bad) if bad else True
    except Exception as e:       # the parent sees what went wrong instead of a bare exit code
        import traceback
        ret[rank] = "exception: %s\n%s" % (e, traceback.format_exc())


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,rw", [("mlperf_like", False), ("mlperf_like", True), ("rmc1_like", True),
                                     ("kaggle_like", False)])
def test_c_hosted_sharded_step_several_ranks_one_gpu(name, rw, world):
    """The C-hosted sharded step with 2 and 3 ranks (whole tables per rank: all-to-all of pooled vectors and their
    gradients with per-peer counts; row-wise tables: reduce-scatter / all-gather; all-reduce of the dense
    gradients) against the single-process oracle.  The ranks share GPU 0 and exchange through dlrm_comm's rehearsal
    transport, which carries out the same calls in shared host memory.  The row-wise cases give the dense gradients
    a communicator of their own (dlrm_model_config.comm_dense: two all-reduces, the top MLP's started early)."""
    import torch.multiprocessing as mp
    from dlrm_amd import comm
    comm_id = comm.unique_id_rehearsal()
    comm_id2 = comm.unique_id_rehearsal() if rw else None
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_rank_worker, args=(world, comm_id, comm_id2, name, rw, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.parametrize("one_id,gemm", [(True, 0), (False, 0), (True, 3)], ids=["one-id", "bags", "one-id-f16x2"])
def test_c_hosted_graphed_steps_equal_direct_launches(one_id, gemm):
    """graph_steps: the step's launches are captured per set of batch buffers and replayed.  Ten steps over three
    resident batches (so every graph is replayed at least twice) leave tables, dense parameters, predictions and loss
    bit-identical to the same steps launched directly; a NULL-stream caller falls back to direct launches."""
    from dlrm_amd.cmodel import CModel
    from dlrm_amd.data import synthetic_batch
    from dlrm_amd.model import DLRMConfig
    cfg = DLRMConfig.kaggle()
    cfg.table_rows = [min(r, 50000) for r in cfg.table_rows]
    B = 512
    dev = torch.device("cuda", 0)
    batches = [synthetic_batch(cfg.table_rows, B, 1, cfg.dense_in, dev, seed=300 + i, loss=cfg.loss, dist="uniform",
                               local_batch=B) for i in range(3)]

    def run(stream, **kw):
        with torch.cuda.stream(stream):
            m = CModel(cfg, B, seed=1, one_id_per_bag=one_id, gemm_mode=gemm, **kw)   # (3 = DLRM_GEMM_F16X2: the passes that
            m.init_dense(2)                                                            #  measure operands are captured too)
            losses = []
            for i in range(10):
                dense, idx, off, labels = batches[i % 3]
                m.load_batch(dense.contiguous(), labels)
                m.train_step(idx, off)
                losses.append(m.loss.clone())
            torch.cuda.synchronize()
            out = (m.weight.clone(), m.params.clone(), m.pred.clone(), torch.cat(losses))
            assert int(m.err_flag.item()) == 0
            m.close()
        return out

    ref = run(torch.cuda.Stream())
    assert float(ref[3][0]) != float(ref[3][-1])
    for name, stream in (("side stream", torch.cuda.Stream()), ("NULL stream", torch.cuda.default_stream())):
        got = run(stream, graph_steps=True)
        for what, a, b in zip(("tables", "dense parameters", "predictions", "losses"), ref, got):
            assert torch.equal(a, b), "%s: %s differ from the direct launches" % (name, what)


@pytest.mark.parametrize("name", ["kaggle_like", "adagrad_bf16rows_off"])
def test_c_hosted_step_without_a_single_id(oracle, name):
    """A batch whose bags are all empty (n_indices = 0): the pooled vectors are zero, no row changes, the dense side
    steps as the oracle's does.  Then a batch with ids again: the step goes on as usual."""
    from dlrm_amd.cmodel import CModel
    cfg = small_cfgs()[name]
    B, T = 37, cfg.num_tables
    m = CModel(cfg, B, seed=3, idx_dtype=torch.int64)
    m.init_dense(7)
    Ws, bs = m.get_dense()
    got = tables_of(m)
    tables = [got[t][0] for t in range(T)]
    state = None if m.opt_state is None else np.zeros(sum(cfg.table_rows), np.float32)
    n_bottom = len(cfg.bottom)
    rng = np.random.default_rng(5)
    for step in range(2):
        dense, idx, off, labels = make_batch(rng, cfg, B, False)
        if step == 0:
            idx = np.zeros(0, np.int64)
            off = np.zeros(T * B + 1, np.int64)
        li, lo = local_batch_ids(m, cfg, idx, off, B)
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        d_idx = torch.from_numpy(li).cuda() if li.size else torch.zeros(1, dtype=torch.int64, device="cuda")[:0]
        loss = m.train_step(d_idx, torch.from_numpy(lo).cuda())
        torch.cuda.synchronize()
        assert int(m.err_flag.item()) == 0
        before = [t.copy() for t in tables]
        loss_ref, pred_ref = oracle_train_step(oracle, cfg, tables, Ws, bs, dense, idx, off, labels, n_bottom, cfg.lr,
                                               feature_order=m.feature_order, emb_opt_state=state)
        if step == 0:
            for a, b in zip(before, tables):
                assert np.array_equal(a, b)
        assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
        assert rel_err(m.pred.cpu().numpy().reshape(-1), pred_ref) <= 1e-5
        for t, (a, _) in tables_of(m).items():
            if step == 0:
                assert np.array_equal(a, before[t]), t
            assert rel_err(a, tables[t]) <= 1e-5, t
        gW, gb = m.get_dense()
        for a, r in zip(gW + gb, Ws + bs):
            assert rel_err(a, r) <= 1e-5
    m.close()


_FUSE_CASE = """
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from dlrm_amd.cmodel import CModel
from dlrm_amd.model import DLRMConfig
from tests.oracle_model import make_batch
from tests.test_cmodel_gpu import local_batch_ids
out = {}
for tag, last, loss in (("a", 64, "bce"), ("b", 320, "mse"), ("c", 1024, "bce"), ("d", 1028, "bce")):
    cfg = DLRMConfig(table_rows=[50, 300, 7], dim=16, dense_in=13, bottom=[32, 16], top=[48, last, 1], lr=0.01, loss=loss,
                     final_relu=loss == "mse")
    B = 1037
    m = CModel(cfg, B, seed=5, idx_dtype=torch.int64)
    m.init_dense(11)
    rng = np.random.default_rng(23)
    for step in range(2):
        dense, idx, off, labels = make_batch(rng, cfg, B, False)
        m.load_batch(torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        li, lo = local_batch_ids(m, cfg, idx, off, B)
        loss_v = m.train_step(torch.from_numpy(li).cuda(), torch.from_numpy(lo).cuda())
        torch.cuda.synchronize()
        assert int(m.err_flag.item()) == 0
        out[tag + "loss%%d" %% step] = loss_v.cpu().clone()
        out[tag + "pred%%d" %% step] = m.pred.cpu().clone()
    Ws, bs = m.get_dense()
    for i, a in enumerate(Ws + bs):
        out[tag + "p%%d" %% i] = torch.from_numpy(np.ascontiguousarray(a))
    m.close()
torch.save(out, sys.argv[1])
"""


def test_c_hosted_fused_last_layer_equals_the_three_launches(tmp_path):
    """A training step evaluates the one-column last layer, the loss gradient and that layer's backward-data pass in
    one launch (up to 1024 inputs; 1028 keeps the separate kernels).  DLRM_FUSE_LAST=0 (read when the model is made:
    two child processes) launches the three kernels: predictions, losses and every dense parameter after two steps
    carry the same bits, with both losses, with and without the output ReLU, on a batch that is no multiple of 16."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "case.py"
    script.write_text(_FUSE_CASE % {"root": root})
    outs = []
    for name, env in (("fused", {}), ("apart", {"DLRM_FUSE_LAST": "0"})):
        out = tmp_path / (name + ".pt")
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=dict(os.environ, **env), timeout=280)
        outs.append(torch.load(out, weights_only=True))
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) > 20
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
