"""MLP GEMMs, dot interaction and losses: HIP kernels vs the CPU oracle (double accumulation).

Every test of the layer products runs in the three evaluations of the float32 products (float32 MFMA chain; three
bf16 pieces, six cross terms; two f16 pieces, three cross terms), see the gemm_mode fixture.

Tolerance: 1e-5 relative to the largest |reference| value of each tensor, and element by element
|err| <= 1e-5 |ref| + 8 * 2^-24 * sum_k |a_ik b_kj| (test_linear_elementwise_error; operands with a wide dynamic range:
tests/test_dense_range_gpu.py).  All three accumulate in float32; for the deepest reductions used here (K = 1024
forward, batch = 8192 in the weight gradient) that rounding stays below the bound."""
import numpy as np
import pytest
import torch

from tests.test_embbag_forward_gpu import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 2], ids=["f32mfma", "bf16x3", "f16x2"], autouse=True)
def gemm_mode(request):
    """every test of this file runs with the three evaluations of the float32 products (f16x2 here without operand
    maxima in the descriptors: the call measures them itself)"""
    from dlrm_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(old)

SHAPES = [
    # M, N, K
    (60, 128, 256),     # reference RMC1 bottom layer 1 (batch 60)
    (60, 32, 128),
    (60, 32, 1056),
    (60, 1, 32),
    (257, 512, 13),     # Criteo bottom layer 1: K = 13
    (300, 1024, 479),   # MLPerf top layer 1: K = 479 (odd)
    (2048, 256, 512),
    (130, 130, 130),    # nothing a multiple of the tile
    (1, 7, 5),
    (9000, 512, 13),    # layers that are no matrix product: a few input columns ...
    (333, 64, 16),
    (9000, 1, 256),     # ... or a single output column (several workgroups' partial sums in the weight gradient)
    (70, 1, 1024),
    (45, 1, 12),
]


def padded(a, mult=4):
    """copy of a 2-D array into a buffer whose row stride is a multiple of 4 floats"""
    r, c = a.shape
    ld = (c + mult - 1) // mult * mult
    buf = torch.zeros((r, ld), dtype=torch.float32, device="cuda")
    buf[:, :c] = torch.from_numpy(a).cuda()
    return buf[:, :c]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_linear_forward(oracle, M, N, K, relu):
    from dlrm_amd import ops
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = oracle.linear_forward(x, w, b, relu)
    y = ops.linear_forward(padded(x), padded(w), torch.from_numpy(b).cuda(), relu)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref) <= 1e-5


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_linear_backward(oracle, M, N, K, relu):
    from dlrm_amd import ops
    rng = np.random.default_rng(M * 5 + N * 11 + K)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)  # x is a ReLU output
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    y = oracle.linear_forward(x, w, b, relu)
    dx_ref, dw_ref, db_ref = oracle.linear_backward(x, w, y, dy, relu)
    dz = dy * (y > 0) if relu else dy
    # the oracle's dx is w.r.t. x; the kernel can additionally apply the ReLU mask of the layer below
    dxm_ref = dx_ref * (x > 0)
    d_dz, d_x, d_w = padded(dz.astype(np.float32)), padded(x), padded(w)
    dx = ops.linear_backward_data(d_dz, d_w)
    dxm = ops.linear_backward_data(d_dz, d_w, relu_mask=d_x)
    dw, db, _ = ops.linear_backward_weight(d_dz, d_x)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5
    assert rel_err(dxm.cpu().numpy(), dxm_ref) <= 1e-5
    assert rel_err(dw.cpu().numpy(), dw_ref) <= 1e-5
    assert rel_err(db.cpu().numpy(), db_ref) <= 1e-5


EPS32 = 2.0 ** -24


def elementwise_excess(got, ref, scale):
    """max over elements of (|got - ref| - 1e-5 |ref|) / (2^-24 * scale): how many units of float32 rounding of
    the element's own absolute sum  scale_ij = sum_k |a_ik b_kj|  the error exceeds 1e-5 relative by."""
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64)) - 1e-5 * np.abs(ref.astype(np.float64))
    return float(np.max(err / (EPS32 * np.maximum(scale, 1e-300))))


@pytest.mark.parametrize("M,N,K", [(2048, 512, 367), (2048, 1024, 479), (4096, 1024, 1024), (300, 256, 512),
                                   (8192, 256, 512)])
def test_linear_elementwise_error(oracle, M, N, K, gemm_mode):
    """Every ELEMENT of the three products of a layer, not only the tensor's norm: |got - ref| <= 1e-5 |ref| +
    8 * 2^-24 * sum_k |a_ik b_kj|.  The second term is the rounding of a float32 accumulation measured against the
    element's own absolute sum (a cancelled dot product cannot be asked for 1e-5 of its value: float32
    accumulation itself, on the CPU reference too, is only good to a few 2^-24 of that sum).  Both evaluations
    (float32 MFMA chain; three-way bfloat16 split, six cross terms) must stay inside it."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dz = rng.standard_normal((M, N)).astype(np.float32)
    y_ref = oracle.linear_forward(x, w, b, False)
    dx_ref, dw_ref, db_ref = oracle.linear_backward(x, w, y_ref, dz, False)
    ax, aw, adz = np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), np.abs(dz).astype(np.float64)
    d_x, d_w, d_dz = padded(x), padded(w), padded(dz)
    y = ops.linear_forward(d_x, d_w, torch.from_numpy(b).cuda(), False)
    dx = ops.linear_backward_data(d_dz, d_w)
    dw, db, _ = ops.linear_backward_weight(d_dz, d_x)
    torch.cuda.synchronize()
    worst = {
        "y": elementwise_excess(y.cpu().numpy(), y_ref, ax @ aw.T + np.abs(b)[None, :]),
        "dx": elementwise_excess(dx.cpu().numpy(), dx_ref, adz @ aw),
        "dw": elementwise_excess(dw.cpu().numpy(), dw_ref, adz.T @ ax),
        "db": elementwise_excess(db.cpu().numpy(), db_ref, adz.sum(0)),
    }
    print("gemm mode %d  M=%d N=%d K=%d  excess in units of 2^-24*abs-sum: %s" % (gemm_mode, M, N, K, worst))
    assert max(worst.values()) <= 8.0, worst


@pytest.mark.parametrize("M,N,K", [(32805, 300, 479), (65536, 128, 256)])
def test_linear_forward_many_tiles(oracle, M, N, K):
    """Products with hundreds of output tiles (the batch sizes of the benchmarks): ragged rows and columns, a
    partial last reduction step read up to the padded leading dimension, bias + ReLU; every element inside
    |err| <= 1e-5 |ref| + 8 * 2^-24 * sum_k |a_ik b_kj| as in test_linear_elementwise_error."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y_ref = oracle.linear_forward(x, w, b, True)
    d_x, d_w = padded(x, 128), padded(w, 128)
    y = ops.linear_forward(d_x, d_w, torch.from_numpy(b).cuda(), True, out=padded(np.zeros((M, N), np.float32), 128))
    torch.cuda.synchronize()
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)[None, :]
    assert rel_err(y.cpu().numpy(), y_ref) <= 1e-5
    assert elementwise_excess(y.cpu().numpy(), y_ref, scale) <= 8.0


def test_backward_weight_large_batch_split(oracle):
    """batch large enough that the reduction is split over workgroups"""
    from dlrm_amd import ops
    rng = np.random.default_rng(3)
    M, N, K = 8192, 64, 96
    x = rng.standard_normal((M, K)).astype(np.float32)
    dz = rng.standard_normal((M, N)).astype(np.float32)
    w = np.zeros((N, K), np.float32)
    _, dw_ref, db_ref = oracle.linear_backward(x, w, np.ones((M, N), np.float32), dz, False, need_dx=False)
    dw, db, _ = ops.linear_backward_weight(padded(dz), padded(x))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().numpy(), dw_ref) <= 1e-5
    assert rel_err(db.cpu().numpy(), db_ref) <= 1e-5


@pytest.mark.parametrize("B,F,D", [(300, 27, 128), (2048, 27, 16), (77, 4, 64), (9, 32, 8)])
def test_interact_dot_views_equal_the_dense_block(B, F, D):
    """Features scattered over several buffers (the per-peer blocks of an exchange, different row strides): the
    interaction over views gives what the kernels give on the assembled [B, F, D] block, bit for bit, forward and
    backward -- also when the gradients are written over the features they were computed from."""
    from dlrm_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B + F + D)
    x = torch.randn((B, F, D), device="cuda", generator=g)
    x[:, 0, :].clamp_(min=0)
    n_out = D + F * (F - 1) // 2
    out_ld = (n_out + 3) // 4 * 4
    dout = torch.randn((B, out_ld), device="cuda", generator=g)
    ref_out = ops.interact_dot_forward(x, out_ld=out_ld)
    ref_dx = ops.interact_dot_backward(x, dout, relu_mask_feature0=True)
    # feature 0 in a wide buffer of its own, the others in blocks of 1, 2, 3, ... features with their own strides
    blocks, f = [], 1
    wide = torch.zeros((B, F * D + 8), device="cuda")
    wide[:, :D] = x[:, 0, :]
    xs = [wide[:, :D]]
    n = 1
    while f < F:
        cnt = min(n, F - f)
        blk = x[:, f:f + cnt, :].reshape(B, cnt * D).contiguous()
        blocks.append(blk)
        xs += [blk[:, i * D:(i + 1) * D] for i in range(cnt)]
        f += cnt
        n += 1
    out = ops.interact_dot_forward_views(xs, out_ld=out_ld)
    dxs = [torch.empty_like(t) for t in xs]
    ops.interact_dot_backward_views(xs, dout, dxs, relu_mask_feature0=True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    for i in range(F):
        assert torch.equal(dxs[i], ref_dx[:, i, :]), i
    ops.interact_dot_backward_views(xs, dout, xs, relu_mask_feature0=True)      # in place
    torch.cuda.synchronize()
    for i in range(F):
        assert torch.equal(xs[i], ref_dx[:, i, :]), i


def test_copy_2d_many():
    """Several strided block copies in one launch (the per-peer blocks of an exchange): every block lands where the
    one-at-a-time copy puts it, untouched elements stay untouched; aligned (16-byte) and unaligned shapes."""
    from dlrm_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for cols_list, ld in (([128, 384, 256, 512, 128], 1408), ([3, 10, 1], 17)):
        B = 37
        src = [torch.randn((B, c), device="cuda", generator=g) for c in cols_list]
        dst = torch.full((B, ld), -7.0, device="cuda")
        ref = dst.clone()
        pairs, c0 = [], 0
        for x in src:
            pairs.append((dst[:, c0:c0 + x.shape[1]], x))
            ref[:, c0:c0 + x.shape[1]] = x
            c0 += x.shape[1]
        pairs = pairs * 4 if len(cols_list) == 3 else pairs       # more than 16 items: several launches
        ops.copy_2d_many_(pairs)
        torch.cuda.synchronize()
        assert torch.equal(dst, ref)


def test_sgd_step(oracle):
    from dlrm_amd import ops
    rng = np.random.default_rng(4)
    w = rng.standard_normal(10007).astype(np.float32)
    g = rng.standard_normal(10007).astype(np.float32)
    dw = torch.from_numpy(w).cuda()
    ops.sgd_step_(dw, torch.from_numpy(g).cuda(), 0.01)
    oracle.sgd_step(w, g, 0.01)
    assert rel_err(dw.cpu().numpy(), w) <= 1e-6


@pytest.mark.parametrize("B,F,D", [(5, 27, 128), (64, 27, 16), (3, 4, 8), (130, 32, 64), (7, 1, 32), (9, 2, 256)])
def test_interact_dot(oracle, B, F, D):
    from dlrm_amd import ops
    rng = np.random.default_rng(B + F + D)
    x = rng.standard_normal((B, F, D)).astype(np.float32)
    x[:, 0, :] = np.maximum(x[:, 0, :], 0)
    ref = oracle.interact_dot_forward(x)
    n_out = ref.shape[1]
    ld = (n_out + 3) // 4 * 4 + 4
    dxd = torch.from_numpy(x).cuda()
    out = ops.interact_dot_forward(dxd, out_ld=ld)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert rel_err(got[:, :n_out], ref) <= 1e-5
    assert (got[:, n_out:] == 0).all()
    dout = rng.standard_normal((B, n_out)).astype(np.float32)
    dref = oracle.interact_dot_backward(x, dout)
    d_dout = torch.zeros((B, ld), device="cuda")
    d_dout[:, :n_out] = torch.from_numpy(dout).cuda()
    dx = ops.interact_dot_backward(dxd, d_dout[:, :n_out])
    dxm = ops.interact_dot_backward(dxd, d_dout[:, :n_out], relu_mask_feature0=True)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dref) <= 1e-5
    dref_m = dref.copy()
    dref_m[:, 0, :] *= (x[:, 0, :] > 0)
    assert rel_err(dxm.cpu().numpy(), dref_m) <= 1e-5


@pytest.mark.parametrize("B,T,D,idx_dtype", [(300, 26, 128, torch.int32), (2048, 26, 16, torch.int32), (77, 3, 64, torch.int64),
                                             (4100, 7, 32, torch.int32), (65, 31, 8, torch.int32),
                                             (131105, 2, 8, torch.int32),     # more chunks of 32 samples than workgroups
                                             (65536, 26, 128, torch.int32)])  # the MLPerf shape at its full batch (kernel against kernel)
def test_interact_dot_gather_equals_pool_then_interact(oracle, B, T, D, idx_dtype):
    """One id per bag: the interaction forward that reads the rows straight from the tables gives, bit for bit, what
    EmbeddingBag forward + the interaction on its output give (same arithmetic on the same values), and its optional
    copy of the gathered block equals the pooled block; the output matches the oracle to 1e-5.  The backward kernel that
    gathers the rows again gives, bit for bit, the gradient of the backward kernel that is handed the block.  A bag that is not one id and an id outside its table
    raise the error flag (2 / 1) and count as zero rows."""
    from dlrm_amd import ops
    rng = np.random.default_rng(B + T + D)
    rows = [int(r) for r in rng.integers(1, 500, size=T)]
    tro = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    weight = torch.from_numpy(rng.standard_normal((int(tro[-1]), D)).astype(np.float32)).cuda()
    idx = np.concatenate([rng.integers(0, rows[t], size=B) for t in range(T)])
    indices = torch.from_numpy(idx).to(idx_dtype).cuda()
    offsets = torch.arange(T * B + 1, dtype=idx_dtype, device="cuda")
    d_tro = torch.from_numpy(tro).cuda()
    F = T + 1
    n_out = D + F * (F - 1) // 2
    out_ld = (n_out + 3) // 4 * 4
    x = torch.zeros((B, F, D), dtype=torch.float32, device="cuda")
    x[:, 0, :] = torch.from_numpy(np.maximum(rng.standard_normal((B, D)), 0).astype(np.float32)).cuda()
    xv = x.view(B, F * D)
    ops.embbag_forward(weight, indices, offsets, B, T, d_tro, include_last_offset=True, out=xv[:, D:], out_ld=F * D)
    ref_out = ops.interact_dot_forward(x, out_ld=out_ld)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    x0 = xv[:, :D]
    out = ops.interact_dot_forward_gather(weight, indices, offsets, B, T, d_tro, x0, out_ld=out_ld, error_flag=flag,
                                          include_last_offset=True)
    # ... and with a copy of the gathered block, written in place around feature 0 (the hosted model's arrangement)
    x2 = torch.zeros_like(x)
    x2[:, 0, :] = x[:, 0, :]
    out2 = ops.interact_dot_forward_gather(weight, indices, offsets, B, T, d_tro, x2.view(B, F * D)[:, :D], out_ld=out_ld,
                                           error_flag=flag, include_last_offset=True, x_copy=x2)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert torch.equal(out, ref_out)
    assert torch.equal(out2, ref_out) and torch.equal(x2, x)
    if B * F * D <= 1 << 24:       # (the full-batch case is held to the other kernels only: bit for bit, above and below)
        z_ref = oracle.interact_dot_forward(x.cpu().numpy())
        assert rel_err(out[:, :n_out].cpu().numpy(), z_ref) <= 1e-5
    # the backward pass that gathers the rows again (no copy of the block) writes what the one given the block writes
    dout = torch.from_numpy(rng.standard_normal((B, out_ld)).astype(np.float32)).cuda()
    for mask in (False, True):
        dx_ref = ops.interact_dot_backward(x, dout, relu_mask_feature0=mask)
        dx = ops.interact_dot_backward_gather(weight, indices, offsets, B, T, d_tro, x0, dout, relu_mask_feature0=mask,
                                              error_flag=flag, include_last_offset=True)
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        assert torch.equal(dx, dx_ref)
    # violations: an id outside its table -> flag 1; a bag that is not exactly one id -> flag 2; both count as zero rows
    bad = indices.clone()
    bad[B // 2] = rows[0] + 5
    x_bad = x.clone()
    x_bad[B // 2, 1, :] = 0
    out_bad = ops.interact_dot_forward_gather(weight, bad, offsets, B, T, d_tro, x0, out_ld=out_ld, error_flag=flag,
                                              include_last_offset=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    assert torch.equal(out_bad, ops.interact_dot_forward(x_bad, out_ld=out_ld))
    flag.zero_()
    dx_bad = ops.interact_dot_backward_gather(weight, bad, offsets, B, T, d_tro, x0, dout, relu_mask_feature0=True,
                                              error_flag=flag, include_last_offset=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    assert torch.equal(dx_bad, ops.interact_dot_backward(x_bad, dout, relu_mask_feature0=True))
    flag.zero_()
    off2 = offsets.clone()
    off2[(T - 1) * B + 3] -= 1
    ops.interact_dot_forward_gather(weight, indices, off2, B, T, d_tro, x0, out_ld=out_ld, error_flag=flag,
                                    include_last_offset=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 2


def test_interact_dot_gather_rejects_what_it_cannot_read():
    """The gathering interaction kernels take float32 rows of a power-of-two width, one id per bag for every table and a
    gradient / output row long enough for the pairs: anything else is an error with a message, not a launch."""
    from dlrm_amd import _capi, ops
    B, T = 64, 3
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    tro = torch.tensor([0, 10, 20, 30], dtype=torch.int64, device="cuda")
    indices = torch.zeros(T * B, dtype=torch.int32, device="cuda")
    offsets = torch.arange(T * B + 1, dtype=torch.int32, device="cuda")

    def both(weight, D, idx=indices, out_ld=None, match=""):
        x0 = torch.zeros((B, D), dtype=torch.float32, device="cuda")
        n_out = D + (T + 1) * T // 2
        dout = torch.zeros((B, n_out if out_ld is None else out_ld), dtype=torch.float32, device="cuda")
        with pytest.raises(_capi.DlrmError, match=match):
            ops.interact_dot_forward_gather(weight, idx, offsets, B, T, tro, x0, out_ld=out_ld, error_flag=flag,
                                            include_last_offset=True)
        with pytest.raises(_capi.DlrmError, match=match):
            ops.interact_dot_backward_gather(weight, idx, offsets, B, T, tro, x0, dout, error_flag=flag,
                                             include_last_offset=True)

    both(torch.zeros((30, 16), dtype=torch.bfloat16, device="cuda"), 16, match="float32 rows")
    both(torch.zeros((30, 24), dtype=torch.float32, device="cuda"), 24, match="power-of-two")
    both(torch.zeros((30, 16), dtype=torch.float32, device="cuda"), 16, idx=indices[:T * B - 1], match="one id per bag")
    both(torch.zeros((30, 16), dtype=torch.float32, device="cuda"), 16, out_ld=16, match="too small")
    torch.cuda.synchronize()
    assert int(flag.item()) == 0


@pytest.mark.parametrize("n", [1, 60, 2048, 65536])
def test_losses(oracle, n):
    from dlrm_amd import ops
    rng = np.random.default_rng(n)
    pred = (rng.standard_normal(n) * 3).astype(np.float32)
    tgt = rng.standard_normal(n).astype(np.float32)
    lab = (rng.random(n) < 0.3).astype(np.float32)
    l_ref, d_ref = oracle.mse_loss(pred, tgt)
    l, d = ops.loss_mse(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda())
    assert abs(l.item() - l_ref) <= 1e-5 * abs(l_ref)
    assert rel_err(d.cpu().numpy(), d_ref) <= 1e-5
    _, dm = ops.loss_mse(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), relu_mask_pred=True)
    assert rel_err(dm.cpu().numpy(), d_ref * (pred > 0)) <= 1e-5
    l_ref, d_ref = oracle.bce_logits_loss(pred, lab)
    l, d = ops.loss_bce_logits(torch.from_numpy(pred).cuda(), torch.from_numpy(lab).cuda())
    assert abs(l.item() - l_ref) <= 1e-5 * abs(l_ref)
    assert rel_err(d.cpu().numpy(), d_ref) <= 1e-5
    # either output on its own (a host asks for the value on a side stream): the same numbers bit for bit
    l1, none = ops.loss_bce_logits(torch.from_numpy(pred).cuda(), torch.from_numpy(lab).cuda(), want_grad=False)
    none2, d1 = ops.loss_bce_logits(torch.from_numpy(pred).cuda(), torch.from_numpy(lab).cuda(), want_loss=False)
    assert none is None and none2 is None and torch.equal(l1, l) and torch.equal(d1, d)
    lm, dmm = ops.loss_mse(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), relu_mask_pred=True)
    lm1, _ = ops.loss_mse(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), relu_mask_pred=True, want_grad=False)
    _, dm1 = ops.loss_mse(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), relu_mask_pred=True, want_loss=False)
    assert torch.equal(lm1, lm) and torch.equal(dm1, dmm)


@pytest.mark.parametrize("rows,cols,ld", [(1, 1, 1), (7, 5, 5), (300, 479, 480), (2048, 512, 512), (4097, 128, 3456),
                                          (1000, 13, 16), (65536, 256, 256)])
def test_absmax_is_the_largest_magnitude(rows, cols, ld, gemm_mode):
    from dlrm_amd import ops
    if gemm_mode != 2:
        pytest.skip("one run is enough")
    gen = torch.Generator(device="cuda").manual_seed(rows + cols)
    buf = torch.randn(rows, ld, device="cuda", generator=gen) * 3.0
    buf[:, cols:] = 1e9                      # what lies between cols and ld is not part of the matrix
    x = buf[:, :cols]
    x[rows // 2, cols // 3] = -77.5          # a negative extreme
    slot = torch.zeros(1, device="cuda")
    ops.absmax_(x, slot)
    assert slot.item() == 77.5
    ops.absmax_(x * 0.5, slot)               # a smaller matrix leaves the slot alone
    assert slot.item() == 77.5
    un = torch.randn(rows * cols + 1, device="cuda", generator=gen)[1:].reshape(rows, cols)   # rows off 16-byte boundaries
    slot.zero_()
    ops.absmax_(un, slot)
    assert slot.item() == un.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(300, 1024, 479), (2048, 256, 512), (130, 130, 130)])
def test_f16x2_with_maxima_handed_in_equals_the_measuring_call(M, N, K, gemm_mode):
    """The operands' maxima only choose the power-of-two scales: handed in through the descriptor (as the hosted model
    does, from the epilogue of the GEMM that wrote the operand) or measured inside the call, the products are the
    same bits; a maximum that is too large by a factor below 2^10 still gives a float32-grade product; and the
    epilogue leaves max |y| behind for the next layer."""
    from dlrm_amd import ops
    if gemm_mode != 2:
        pytest.skip("f16x2 only")
    gen = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = padded(torch.relu(torch.randn(M, K, device="cuda", generator=gen)).cpu().numpy())
    w = padded((torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5).cpu().numpy())
    b = torch.randn(N, device="cuda", generator=gen)
    dz = padded((torch.randn(M, N, device="cuda", generator=gen) * 1e-5).cpu().numpy())
    ax, aw, adz = (t.abs().max().reshape(1).clone() for t in (x, w, dz))
    ay, adx = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    y0 = ops.linear_forward(x, w, b, True)
    y1 = ops.linear_forward(x, w, b, True, x_absmax=ax, w_absmax=aw, y_absmax=ay)
    dx0 = ops.linear_backward_data(dz, w, relu_mask=x)
    dx1 = ops.linear_backward_data(dz, w, relu_mask=x, dz_absmax=adz, w_absmax=aw, dx_absmax=adx)
    dw0, db0, _ = ops.linear_backward_weight(dz, x)
    dw1, db1, _ = ops.linear_backward_weight(dz, x, dz_absmax=adz, x_absmax=ax)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    assert ay.item() == y1.abs().max().item() and adx.item() == dx1.abs().max().item()
    y2 = ops.linear_forward(x, w, b, True, x_absmax=ax * 900.0, w_absmax=aw * 3.0)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert ((y2.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6


@pytest.mark.parametrize("M,N,K", [(8192, 512, 479), (8192, 512, 512), (16421, 256, 1024), (32768, 128, 256),
                                   (32768, 512, 479), (32805, 300, 256)])    # (the last two: 256-column tiles)
@pytest.mark.parametrize("relu", [False, True])
def test_linear_with_presplit_weights(oracle, M, N, K, relu, gemm_mode):
    """f16x2 with the weight matrix cut into its f16 planes beforehand (dlrm_split_weights): the forward and
    backward-data products through the kernel that reads the planes (grids of 256 tiles and more: ragged rows, a
    partial last reduction step, a partial last column tile, bias / ReLU / ReLU mask fused) against the oracle,
    element by element, and against the same products with the split inside the kernel."""
    from dlrm_amd import ops
    if gemm_mode != 2:
        pytest.skip("the planes belong to the f16x2 evaluation")
    rng = np.random.default_rng(M + N + K)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dz = (rng.standard_normal((M, N)) * (rng.random((M, N)) > 0.5)).astype(np.float32)
    y_ref = oracle.linear_forward(x, w, b, relu)
    dx_ref, _, _ = oracle.linear_backward(x, w, np.ones_like(y_ref), dz, False)
    dxm_ref = dx_ref * (x > 0)
    d_x, d_w, d_dz, d_b = padded(x, 32), padded(w), padded(dz), torch.from_numpy(b).cuda()
    sw = ops.SplitWeights(d_w)
    y = ops.linear_forward(d_x, d_w, d_b, relu, w_planes=sw)
    y_in = ops.linear_forward(d_x, d_w, d_b, relu)
    dx = ops.linear_backward_data(d_dz, d_w, w_planes=sw)
    dxm = ops.linear_backward_data(d_dz, d_w, relu_mask=d_x, w_planes=sw)
    dx_in = ops.linear_backward_data(d_dz, d_w)
    torch.cuda.synchronize()
    ax, aw, adz = np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), np.abs(dz).astype(np.float64)
    assert rel_err(y.cpu().numpy(), y_ref) <= 1e-5 and rel_err(dx.cpu().numpy(), dx_ref) <= 1e-5
    assert rel_err(dxm.cpu().numpy(), dxm_ref) <= 1e-5
    assert elementwise_excess(y.cpu().numpy(), y_ref, ax @ aw.T + np.abs(b)[None, :]) <= 8.0
    assert elementwise_excess(dx.cpu().numpy(), dx_ref, adz @ aw) <= 8.0
    # same pieces, same products, another order of the float32 accumulation: a few ulps of the absolute sum apart
    assert elementwise_excess(y.cpu().numpy(), y_in.cpu().numpy().astype(np.float64), ax @ aw.T + np.abs(b)[None, :]) <= 8.0
    assert elementwise_excess(dx.cpu().numpy(), dx_in.cpu().numpy().astype(np.float64), adz @ aw) <= 8.0


_SHAPE_CASE = """
import sys
sys.path.insert(0, %(root)r)
import torch
from dlrm_amd import ops
ops.set_gemm_mode(2)
g = torch.Generator(device="cuda").manual_seed(3)
out = {}
for M, N, K in ((16421, 256, 1024), (32805, 300, 479)):
    Kp = (K + 31) // 32 * 32
    x = torch.relu(torch.randn(M, Kp, device="cuda", generator=g))[:, :K]
    w = (torch.randn(N, Kp, device="cuda", generator=g) / K ** 0.5)[:, :K]
    b = torch.randn(N, device="cuda", generator=g)
    dz = torch.randn(M, N, device="cuda", generator=g) * (torch.rand(M, N, device="cuda", generator=g) > 0.5)
    sw = ops.SplitWeights(w)
    out["y%%d" %% K] = ops.linear_forward(x, w, b, True, w_planes=sw).cpu()
    out["dx%%d" %% K] = ops.linear_backward_data(dz, w, relu_mask=x, w_planes=sw).cpu()
    out["ref_y%%d" %% K] = torch.relu(x.double() @ w.double().t() + b.double()).cpu()
    out["ref_dx%%d" %% K] = ((dz.double() @ w.double()) * (x > 0)).cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
"""


def test_both_mfma_shapes_of_the_planes_kernel(tmp_path):
    """The kernel that reads pre-split weights exists on v_mfma_f32_16x16x32_f16 (the default) and on 32x32x16
    (DLRM_F16P_SHAPE=32, read once per process: two child processes).  Same pieces, same products, another order of the
    float32 accumulation: both must meet the float64 product at the tolerance of the other tests, on a ragged batch, a
    partial last reduction step and a partial last column tile."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "case.py"
    script.write_text(_SHAPE_CASE % {"root": root})
    outs = []
    for name, env in (("s16", {}), ("s32", {"DLRM_F16P_SHAPE": "32"})):
        out = tmp_path / (name + ".pt")
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=dict(os.environ, **env), timeout=280)
        outs.append(torch.load(out, weights_only=True))
    for o in outs:
        for k in ("y1024", "dx1024", "y479", "dx479"):
            ref = o["ref_" + k]
            assert ((o[k].double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6, k
    for k in ("y1024", "dx1024", "y479", "dx479"):
        assert ((outs[0][k].double() - outs[1][k].double()).abs().max() / outs[0]["ref_" + k].abs().max()).item() <= 2e-6
