"""The three products of a layer on operands with a wide dynamic range, in all three evaluations.

float32 arithmetic (the reference: torch::nn::Linear on float32, dlrm_libtorch/src/dlrmtest.cpp:254-272) rounds every
element relative to ITSELF, whatever the rest of the tensor holds.  An evaluation that scales a whole operand by one
power of two and cuts it into 16-bit pieces does not do that by construction, so it is held here, element by element,
to the bound the N(0,1) tests use,

    |got - ref| <= 1e-5 |ref| + 8 * 2^-24 * sum_k |a_ik b_kj|,

on operands where one scale per tensor is hard: a sample (row) 10^4 and 10^7 times larger than the others, an input
feature (column) like that, gradients drawn log-normally over more than 12 decades with ReLU zeros among them, a weight
matrix with one large column or row, a block far below everything else.

What the f16x2 evaluation guarantees (common.h: split2_pair): every element within 2^27 of its operand's largest
magnitude carries 22 mantissa bits; anything smaller has an absolute error of at most 2^-49 of that largest magnitude.
Whole rows more than 2^27 below the tensor's maximum are therefore outside the float32-equivalent range; for those
the last test asserts the guaranteed floor itself (and that the bf16x3 and float32-MFMA evaluations need no floor)."""
import numpy as np
import pytest
import torch

from tests.test_dense_gpu import elementwise_excess, padded

pytestmark = pytest.mark.gpu

N, K = 384, 479
M = 1024            # rebound per test: the kernel that reads pre-split weights only takes grids of 256 tiles and more


@pytest.fixture(params=[0, 1, 2, 3], ids=["f32mfma", "bf16x3", "f16x2", "f16x2-planes"], autouse=True)
def gemm_mode(request):
    """3: f16x2 with the weights split beforehand (dlrm_split_weights), the forward and backward-data products through
    the kernel that reads those planes -- on a batch large enough for that kernel to be chosen"""
    global M
    from dlrm_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(min(request.param, 2))
    M = 11008 if request.param == 3 else 1024
    yield request.param
    ops.set_gemm_mode(old)
    M = 1024


def base_operands(seed):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    dz = (rng.standard_normal((M, N)) * (rng.random((M, N)) > 0.5)).astype(np.float32)    # ReLU zeros
    return rng, x, w, dz


def make_case(name):
    rng, x, w, dz = base_operands(sum(map(ord, name)))
    if name.startswith("sample_"):             # one sample far larger than the rest, in the activations and in the gradients
        f = float(name.split("_")[1])
        x[7] *= f
        dz[7] *= f
    elif name.startswith("feature_"):          # one input feature / one unit's gradient far larger than the rest
        f = float(name.split("_")[1])
        x[:, 5] *= f
        dz[:, 9] *= f
    elif name == "dz_lognormal":               # +-6 decades at three sigma; ReLU zeros stay
        dz = (np.exp(rng.standard_normal((M, N)) * 2 * np.log(10)) * np.sign(dz)).astype(np.float32)
    elif name == "x_loguniform":               # activations uniform in the exponent over 12 decades
        x = (10.0 ** rng.uniform(-6, 6, (M, K)) * (x > 0)).astype(np.float32)
    elif name == "w_column_1e6":
        w[:, 11] *= 1e6
    elif name == "w_row_1e6":
        w[3] *= 1e6
    elif name == "block_1e-13":                # a 128 x 128 block that becomes zero in 16-bit pieces under one scale per tensor
        x[100:228, 100:228] *= 1e-13
        dz[300:428, 60:188] *= 1e-13
    elif name == "denormal_inputs":            # float32 subnormals among ordinary values
        x[::7, ::5] = 1e-41
        dz[::3, ::11] = -3e-42
    else:
        raise KeyError(name)
    return x, w, dz


def products(oracle, x, w, dz, planes=False):
    from dlrm_amd import ops
    b = np.zeros(N, np.float32)
    y_ref = oracle.linear_forward(x, w, b, False)
    dx_ref, dw_ref, db_ref = oracle.linear_backward(x, w, y_ref, dz, False)
    d_x, d_w, d_dz = padded(x, 32), padded(w), padded(dz)
    kw = dict(w_planes=ops.SplitWeights(d_w)) if planes else {}
    y = ops.linear_forward(d_x, d_w, torch.from_numpy(b).cuda(), False, **kw)
    dx = ops.linear_backward_data(d_dz, d_w, **kw)
    dw, db, _ = ops.linear_backward_weight(d_dz, d_x)
    torch.cuda.synchronize()
    got = {"y": y.cpu().numpy(), "dx": dx.cpu().numpy(), "dw": dw.cpu().numpy(), "db": db.cpu().numpy()}
    ref = {"y": y_ref, "dx": dx_ref, "dw": dw_ref, "db": db_ref}
    ax, aw, adz = np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), np.abs(dz).astype(np.float64)
    scale = {"y": ax @ aw.T, "dx": adz @ aw, "dw": adz.T @ ax, "db": adz.sum(0)}
    return got, ref, scale, (ax, aw, adz)


CASES = ["sample_1e4", "sample_1e7", "feature_1e4", "feature_1e7", "dz_lognormal", "x_loguniform", "w_column_1e6",
         "w_row_1e6", "block_1e-13", "denormal_inputs"]


@pytest.mark.parametrize("case", CASES)
def test_wide_range_operands_elementwise(oracle, case, gemm_mode):
    x, w, dz = make_case(case)
    got, ref, scale, _ = products(oracle, x, w, dz, planes=gemm_mode == 3)
    worst = {k: elementwise_excess(got[k], ref[k], scale[k]) for k in got}
    print("gemm mode %d  %-16s excess in units of 2^-24*abs-sum: %s" % (gemm_mode, case, worst))
    assert all(np.isfinite(got[k]).all() for k in got)
    assert max(worst.values()) <= 8.0, worst


@pytest.mark.parametrize("factor", [1e-9, 1e-13, 1e-30])
def test_rows_far_below_the_tensor_maximum(oracle, factor, gemm_mode):
    """128 whole samples `factor` times smaller than the rest.  The float32 MFMA and the exact three-way bf16 split keep
    the element-wise bound; two f16 pieces under one scale per tensor keep it with the guaranteed floor added,
    2^-48 * (max|a| * sum_k |b_kj| + max|b| * sum_k |a_ik|)."""
    _, x, w, dz = base_operands(int(-np.log10(factor)))
    x[100:228] *= factor
    dz[100:228] *= factor
    got, ref, scale, (ax, aw, adz) = products(oracle, x, w, dz, planes=gemm_mode == 3)
    if gemm_mode >= 2:
        fl = 2.0 ** -48 / 2.0 ** -24 / 8.0      # the floor in the units elementwise_excess divides by, for a limit of 8
        scale["y"] = scale["y"] + fl * (ax.max() * aw.sum(1)[None, :] + aw.max() * ax.sum(1)[:, None])
        scale["dx"] = scale["dx"] + fl * (adz.max() * aw.sum(0)[None, :] + aw.max() * adz.sum(1)[:, None])
        scale["dw"] = scale["dw"] + fl * (adz.max() * ax.sum(0)[None, :] + ax.max() * adz.sum(0)[:, None])
    worst = {k: elementwise_excess(got[k], ref[k], scale[k]) for k in got}
    print("gemm mode %d  rows x %g  excess: %s" % (gemm_mode, factor, worst))
    assert max(worst.values()) <= 8.0, worst
