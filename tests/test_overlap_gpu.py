"""Kernels of different streams share the compute units; none of them may change another one's results.

The scenario that once failed: the fused sparse row update (segmented sum over the sorted occurrences + SGD) on one
stream while the weight-gradient GEMMs of the top MLP run on a second one -- the C-hosted step does exactly that
whenever the weight-gradient stream lags (timing events on it are enough).  Rows came out with single components of
the update missing in a fraction of the runs; a build without packed float32 VALU instructions does not show it (see
dlrm_amd/csrc/Makefile).  Bar: bit-identical to the serial run, every repetition."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = [1460, 583, 1000000, 200000, 305, 24, 12517, 633, 3, 93145, 5683, 800000, 3194, 27, 14992, 500000, 10, 5652, 2173,
        4, 700000, 18, 15, 286181, 105, 142572]


@pytest.mark.parametrize("gemm", [1, 2], ids=["bf16x3", "f16x2"])
def test_sparse_update_beside_weight_gradients_equals_the_serial_run(gemm):
    from dlrm_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(gemm)
    try:
        _update_beside_weight_gradients(ops)
    finally:
        ops.set_gemm_mode(old)


def _update_beside_weight_gradients(ops):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    B, T, D = 2048, len(ROWS), 16
    tro = torch.tensor([0] + torch.tensor(ROWS).cumsum(0).tolist(), dtype=torch.int64, device=dev)
    w0 = torch.randn(int(tro[-1]), D, device=dev, generator=gen)
    idx = torch.cat([torch.randint(0, r, (B,), device=dev, generator=gen) for r in ROWS]).to(torch.int64)
    off = torch.arange(T * B, device=dev, dtype=torch.int64)
    grad = torch.randn(B, T * D, device=dev, generator=gen) * 1e-3
    layers = []
    for N, K in ((256, 512), (512, 367)):          # the top MLP of the Criteo-Kaggle model
        layers.append((torch.randn(B, N, device=dev, generator=gen), torch.randn(B, K, device=dev, generator=gen),
                       torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
    wsd = torch.empty(max(ops.linear_backward_weight_workspace_bytes(B, dz.shape[1], x.shape[1]) for dz, x, _, _ in layers),
                      dtype=torch.uint8, device=dev)
    ws = ops.embbag_backward_prepare(w0, idx, off, B, num_tables=T, table_row_offsets=tro)

    def apply(w):
        ops.embbag_backward_apply(w, idx, off, B, grad, ws, num_tables=T, table_row_offsets=tro, lr=0.1)

    ref = w0.clone()
    apply(ref)
    dw_ref = []
    for dz, x, dw, db in layers:
        ops.linear_backward_weight(dz, x, dw, db, workspace=wsd)
        dw_ref.append((dw.clone(), db.clone()))
    torch.cuda.synchronize()
    assert not torch.equal(ref, w0)

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(8):
        w = w0.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(8):
                for dz, x, dw, db in layers:
                    ops.linear_backward_weight(dz, x, dw, db, workspace=wsd)
        with torch.cuda.stream(sa):
            apply(w)
        torch.cuda.synchronize()
        bad = torch.nonzero((w != ref).any(dim=1)).flatten()
        assert bad.numel() == 0, "repetition %d: %d rows differ from the serial update, first %s" % (
            rep, bad.numel(), bad[:4].tolist())
        for (dz, x, dw, db), (dwr, dbr) in zip(layers, dw_ref):
            assert torch.equal(dw, dwr) and torch.equal(db, dbr)


def test_f16_piece_products_beside_bf16_piece_weight_gradients_equal_the_serial_run():
    """The other direction: the victims are the kernels whose operand split uses v_cvt_pk_f16_f32 / v_fma_mix_f32 (the
    f16x2 forward and backward-data products, with and without pre-split weights), the neighbours the bf16x3
    weight-gradient GEMMs that disturbed the sparse update."""
    from dlrm_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    M = 8192
    x = torch.relu(torch.randn(M, 512, device=dev, generator=gen))
    w = torch.randn(1024, 512, device=dev, generator=gen) / 512 ** 0.5
    dz = torch.randn(M, 1024, device=dev, generator=gen) * (torch.rand(M, 1024, device=dev, generator=gen) > 0.5)
    bias = torch.randn(1024, device=dev, generator=gen)
    agg = [(torch.randn(M, 256, device=dev, generator=gen), torch.randn(M, 512, device=dev, generator=gen),
            torch.empty(256, 512, device=dev), torch.empty(256, device=dev))]
    wsd = torch.empty(ops.linear_backward_weight_workspace_bytes(M, 256, 512), dtype=torch.uint8, device=dev)
    old = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode(2)
        planes = ops.SplitWeights(w)

        def victims():
            return [ops.linear_forward(x, w, bias, True), ops.linear_backward_data(dz, w, relu_mask=x),
                    ops.linear_forward(x, w, bias, True, w_planes=planes), ops.linear_backward_data(dz, w, w_planes=planes)]

        ref = victims()
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(6):
            torch.cuda.synchronize()
            ops.set_gemm_mode(1)
            with torch.cuda.stream(sb):
                for _ in range(12):
                    for gz, gx, dw, db in agg:
                        ops.linear_backward_weight(gz, gx, dw, db, workspace=wsd)
            ops.set_gemm_mode(2)
            with torch.cuda.stream(sa):
                got = victims()
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(got, ref)):
                assert torch.equal(a, b), "repetition %d, product %d differs from the serial run" % (rep, i)
    finally:
        ops.set_gemm_mode(old)


@pytest.mark.parametrize("shape", ["kaggle", "mlperf_capped"])
def test_c_hosted_step_with_a_lagging_weight_gradient_stream_is_repeatable(shape):
    """Timing events around every weight gradient hold that stream back until it overlaps the sparse update; tables,
    dense parameters and predictions of two steps must not depend on it."""
    from dlrm_amd.cmodel import CModel
    from dlrm_amd.data import synthetic_batch
    from dlrm_amd.model import DLRMConfig
    if shape == "kaggle":
        cfg, B = DLRMConfig.kaggle(), 2048
    else:                                   # the MLPerf layers and row width at the local batch of an 8-GPU run
        cfg, B = DLRMConfig.mlperf(), 8192
        cfg.table_rows = [min(r, 200000) for r in cfg.table_rows]
    dev = torch.device("cuda", 0)
    batches = [synthetic_batch(cfg.table_rows, B, 1, cfg.dense_in, dev, seed=1000 + i, loss=cfg.loss, dist="uniform",
                               local_batch=B) for i in range(2)]

    def run(timed):
        m = CModel(cfg, B, seed=0, one_id_per_bag=True)
        m.init_dense(0)
        if timed:
            m.enable_timing(timed, 4000)
        for dense, idx, off, labels in batches:
            m.load_batch(dense.contiguous(), labels)
            m.train_step(idx, off)
        torch.cuda.synchronize()
        out = (m.weight.clone(), m.params.clone(), m.pred.clone())
        m.close()
        return out

    ref = run(None)
    for rep in range(4):
        got = run(["linear_backward_weight"] if rep % 2 == 0 else "all")
        for name, a, b in zip(("tables", "dense parameters", "predictions"), ref, got):
            assert torch.equal(a, b), "repetition %d: %s differ" % (rep, name)
