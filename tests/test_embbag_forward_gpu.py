"""EmbeddingBag forward: HIP kernel vs the CPU oracle on seeded inputs.

Tolerance: float32 outputs within 1e-5 relative (of the largest |reference|
value of the tensor) -- the bar the hot path is held to against libtorch; bf16
rows are exact inputs, so the same 1e-5 bound applies to their fp32 sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    return float(np.max(np.abs(a - ref)) / max(np.max(np.abs(ref)), 1e-30))


def make_case(rng, T, B, D, rows_list, L, ragged):
    tro = np.concatenate([[0], np.cumsum(rows_list)]).astype(np.int64)
    w = rng.standard_normal((int(tro[-1]), D)).astype(np.float32)
    n_bags = T * B
    if ragged:
        lens = rng.integers(0, 2 * L + 1, size=n_bags)
        lens[rng.integers(0, n_bags, size=max(1, n_bags // 7))] = 0  # empty bags
    else:
        lens = np.full(n_bags, L)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.empty(int(off[-1]), dtype=np.int64)
    for t in range(T):
        s, e = off[t * B], off[(t + 1) * B]
        idx[s:e] = rng.integers(0, rows_list[t], size=e - s)
    return w, tro, idx, off


CASES = [
    # T, B, D, rows, L, ragged
    (1, 7, 16, [50], 1, False),
    (3, 33, 128, [100, 3, 1000], 1, False),
    (26, 64, 128, None, 1, False),
    (2, 40, 32, [500, 20], 80, False),      # reference RMC1 bag shape
    (4, 19, 64, [64, 1, 7, 300], 5, True),
    (2, 9, 256, [30, 40], 3, True),
    (2, 5, 512, [30, 40], 9, True),          # rows wider than one wave pass
    (1, 3, 20, [11], 4, True),               # dim that needs the element-per-lane path
    (2, 130, 64, [1000, 2000], 100, False),  # long bags, few of them: wave-per-bag kernel
    (1, 4, 128, [300], 200, True),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("rows_t", ["f32", "bf16"])
@pytest.mark.parametrize("last", [False, True])
def test_forward_matches_oracle(oracle, case, idx_t, mode, rows_t, last):
    from dlrm_amd import ops
    T, B, D, rows_list, L, ragged = case
    rng = np.random.default_rng(hash((T, B, D, L)) % (1 << 31))
    if rows_list is None:
        rows_list = [int(x) for x in rng.integers(1, 5000, size=T)]
    w, tro, idx, off = make_case(rng, T, B, D, rows_list, L, ragged)
    pooling = oracle.POOL_MEAN if mode == "mean" else oracle.POOL_SUM
    if rows_t == "bf16":
        wb = oracle.f32_to_bf16(w)
        ref = oracle.embbag_forward(wb, idx, off, B, T, tro, pooling, include_last_offset=True)
        dw = torch.from_numpy(wb.view(np.int16)).cuda().view(torch.bfloat16)
    else:
        ref = oracle.embbag_forward(w, idx, off, B, T, tro, pooling, include_last_offset=True)
        dw = torch.from_numpy(w).cuda()
    off_in = off if last else off[:-1]
    d_idx = torch.from_numpy(idx).to(idx_t).cuda()
    d_off = torch.from_numpy(np.ascontiguousarray(off_in)).to(idx_t).cuda()
    d_tro = torch.from_numpy(tro).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = ops.embbag_forward(dw, d_idx, d_off, B, T, d_tro, pooling, include_last_offset=last, error_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert rel_err(out.cpu().numpy(), ref) <= 1e-5


def test_forward_into_interaction_layout_and_bad_index(oracle):
    """Pooled vectors land in a [B, 1+T, D] buffer (slot 0 left for the dense
    vector); an out-of-range index is skipped and reported."""
    from dlrm_amd import ops
    rng = np.random.default_rng(5)
    T, B, D = 3, 17, 64
    w, tro, idx, off = make_case(rng, T, B, D, [40, 50, 60], 2, False)
    ref = oracle.embbag_forward(w, idx, off, B, T, tro, include_last_offset=True)
    buf = torch.full((B, T + 1, D), -7.0, device="cuda")
    ops.embbag_forward(torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(off).cuda(),
                       B, T, torch.from_numpy(tro).cuda(), include_last_offset=True,
                       out=buf.view(-1)[D:], out_ld=(T + 1) * D)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert (got[:, 0, :] == -7.0).all()
    assert rel_err(got[:, 1:, :].reshape(B, T * D), ref) <= 1e-5

    bad = idx.copy()
    bad[3] = 10 ** 6
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = ops.embbag_forward(torch.from_numpy(w).cuda(), torch.from_numpy(bad).cuda(), torch.from_numpy(off).cuda(),
                             B, T, torch.from_numpy(tro).cuda(), include_last_offset=True, error_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    # every other bag is still right
    got = out.cpu().numpy()
    bag = int(np.searchsorted(off, 3, side="right") - 1)
    mask = np.ones((B, T), dtype=bool)
    mask[bag % B, bag // B] = False
    assert np.allclose(got.reshape(B, T, D)[mask], ref.reshape(B, T, D)[mask], rtol=1e-5, atol=1e-6)


def test_table_init_matches_oracle(oracle):
    from dlrm_amd import ops
    w = torch.empty((1000, 32), device="cuda")
    ops.table_init_uniform_(w, seed=1234, first=77)
    ref = oracle.table_init_uniform(w.numel(), 1234, 77).reshape(1000, 32)
    assert np.array_equal(w.cpu().numpy(), ref)
    assert ref.min() >= 0.0 and ref.max() < 1.0
