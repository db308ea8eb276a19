"""Sparse backward + fused row-wise SGD / Adagrad: HIP kernels vs the CPU oracle.

Tolerance: updated float32 rows within 1e-5 relative (of the largest
|reference| weight); bf16 rows within 1e-2 relative (one bf16 rounding of the
updated value, 2^-8).  Untouched rows must be bit-identical."""
import numpy as np
import pytest
import torch

from tests.test_embbag_forward_gpu import make_case, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # T, B, D, rows, L, ragged
    (1, 7, 16, [50], 1, False),
    (3, 33, 128, [100, 3, 1000], 1, False),   # table of 3 rows: heavy duplicates
    (26, 64, 128, None, 1, False),
    (2, 40, 32, [500, 20], 80, False),
    (4, 19, 64, [64, 1, 7, 300], 5, True),
    (2, 5, 512, [30, 40], 9, True),
    (1, 3, 20, [11], 4, True),
    (1, 1000, 64, [2], 1, False),            # two rows: runs of ~500, cut into segments and recombined
    (2, 300, 128, [1, 5], 3, False),         # one row hit 900 times
    (1, 256, 32, [1], 1, False),             # run length an exact multiple of the segment length
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64])
@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("rows_t", ["f32", "bf16"])
@pytest.mark.parametrize("opt", ["sgd", "adagrad", "sgd_last_wins"])
def test_backward_update_matches_oracle(oracle, case, idx_t, mode, rows_t, opt):
    from dlrm_amd import ops
    T, B, D, rows_list, L, ragged = case
    rng = np.random.default_rng(hash((T, B, D, L, 1)) % (1 << 31))
    if rows_list is None:
        rows_list = [int(x) for x in rng.integers(1, 5000, size=T)]
    w, tro, idx, off = make_case(rng, T, B, D, rows_list, L, ragged)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    pooling = oracle.POOL_MEAN if mode == "mean" else oracle.POOL_SUM
    optimizer = oracle.OPT_ADAGRAD_ROWWISE if opt == "adagrad" else oracle.OPT_SGD
    dup = oracle.DUP_LAST_WINS if opt == "sgd_last_wins" else oracle.DUP_ACCUMULATE
    state = (rng.random(w.shape[0]).astype(np.float32) * 0.1) if opt == "adagrad" else None
    lr = 0.05

    if rows_t == "bf16":
        w_ref = oracle.f32_to_bf16(w)
        dw = torch.from_numpy(w_ref.view(np.int16).copy()).cuda().view(torch.bfloat16)
    else:
        w_ref = w.copy()
        dw = torch.from_numpy(w).cuda()
    w_before = w_ref.copy()
    st_ref = None if state is None else state.copy()
    oracle.embbag_backward_update(w_ref, idx, off, B, grad, T, tro, pooling, True, optimizer, dup, lr, 1e-8, st_ref)

    d_state = None if state is None else torch.from_numpy(state).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.embbag_backward_update(dw, torch.from_numpy(idx).to(idx_t).cuda(), torch.from_numpy(off).to(idx_t).cuda(),
                               B, torch.from_numpy(grad).cuda(), T, torch.from_numpy(tro).cuda(), pooling, True,
                               optimizer, dup, lr, 1e-8, d_state, error_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    if rows_t == "bf16":
        got_bits = dw.view(torch.int16).cpu().numpy().view(np.uint16)
        got, ref = oracle.bf16_to_f32(got_bits), oracle.bf16_to_f32(w_ref)
        untouched = (w_ref == w_before).all(axis=1)
        assert np.array_equal(got_bits[untouched], w_before[untouched])
        assert rel_err(got, ref) <= 1e-2
    else:
        got, ref = dw.cpu().numpy(), w_ref
        untouched = (w_ref == w_before).all(axis=1)
        assert np.array_equal(got[untouched], w_before[untouched])
        assert rel_err(got, ref) <= 1e-5
    if state is not None:
        assert rel_err(d_state.cpu().numpy(), st_ref) <= 1e-5


def test_backward_reports_bad_index_and_updates_the_rest(oracle):
    from dlrm_amd import ops
    rng = np.random.default_rng(11)
    T, B, D = 2, 16, 32
    w, tro, idx, off = make_case(rng, T, B, D, [100, 200], 3, False)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    bad = idx.copy()
    bad[5] = -4
    good_mask = np.ones(idx.size, dtype=bool)
    good_mask[5] = False
    # oracle on the same batch with occurrence 5 removed
    lens = np.diff(off)
    bag5 = int(np.searchsorted(off, 5, side="right") - 1)
    lens2 = lens.copy(); lens2[bag5] -= 1
    off2 = np.concatenate([[0], np.cumsum(lens2)])
    w_ref = w.copy()
    oracle.embbag_backward_update(w_ref, idx[good_mask], off2, B, grad, T, tro, include_last_offset=True, lr=0.1)
    dw = torch.from_numpy(w).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.embbag_backward_update(dw, torch.from_numpy(bad).cuda(), torch.from_numpy(off).cuda(), B,
                               torch.from_numpy(grad).cuda(), T, torch.from_numpy(tro).cuda(),
                               include_last_offset=True, lr=0.1, error_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    assert rel_err(dw.cpu().numpy(), w_ref) <= 1e-5


_SORT_CASE = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from dlrm_amd import ops
g = torch.Generator(device="cuda").manual_seed(11)
rows = [3, 70000, 1, 5000000, 129, 2, 40000, 900001]           # 1 .. 23 key bits per table, 23 bits in all: three passes
B, D, L = 6001, 32, 3
tro = torch.tensor([0] + torch.tensor(rows).cumsum(0).tolist(), dtype=torch.int64, device="cuda")
w = torch.randn(int(tro[-1]), D, device="cuda", generator=g)
idx = torch.cat([torch.randint(0, r, (B * L,), device="cuda", generator=g) for r in rows]).to(torch.int32)
idx[5::97] = -1                                                  # ids the update must skip
off = torch.arange(0, len(rows) * B * L + 1, L, device="cuda", dtype=torch.int32)
grad = torch.randn(B, len(rows) * D, device="cuda", generator=g)
for last_wins in (False, True):
    ops.embbag_backward_update(w, idx, off, B, grad, len(rows), tro, 0, True, lr=0.1,
                               dup_mode=1 if last_wins else 0)
torch.cuda.synchronize()
torch.save(w.cpu(), sys.argv[1])
"""


def test_own_sort_gives_the_rows_the_rocprim_sort_gives(tmp_path):
    """The library's own pair sort (pair_sort.hip) against rocPRIM's radix sort (DLRM_SORT=rocprim, read once per
    process: two child processes): the same seeded update -- eight tables of 1 to 23 key bits, three ids per bag,
    out-of-range ids among them, accumulate then last-wins (which depends on the order INSIDE a run of equal keys,
    i.e. on the sort's stability) -- must leave bit-identical tables."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "case.py"
    script.write_text(_SORT_CASE % {"root": root})
    outs = []
    for name, env in (("own", {}), ("rocprim", {"DLRM_SORT": "rocprim"})):
        out = tmp_path / (name + ".pt")
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=dict(os.environ, **env), timeout=280)
        outs.append(torch.load(out, weights_only=True))
    assert torch.equal(outs[0], outs[1])
