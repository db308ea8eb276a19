"""dlrm_linear_backward_weight_static (the entry point that allocates nothing: the split reduction's partial sums
live in a static device array of the library) against dlrm_linear_backward_weight with a caller workspace.
Same kernels and the same split, so the outputs must be bit-identical.  Two calls in a row, the second on a
smaller layer (the array keeps the first call's contents), then a layer beyond the array's size, which must be
refused.  The second test checks the property itself: no allocation, and the call is capturable."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(M, N, K, seed):
    from dlrm_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    dz = torch.randn((M, N), generator=g, device="cuda")
    x = torch.randn((M, K), generator=g, device="cuda")
    dw_ref, db_ref, _ = ops.linear_backward_weight(dz, x)
    dw = torch.full((N, K), float("nan"), device="cuda")
    db = torch.full((N,), float("nan"), device="cuda")
    ops.linear_backward_weight_static(dz, x, dw, db)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref), (M, N, K)
    # and against float64 on the host, so that "equal" is not "equally wrong"
    ref = dz.double().t() @ x.double()
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.parametrize("mode", [1, 0, 2], ids=["bf16x3", "f32mfma", "f16x2"])
def test_static_scratch_entry_point_matches_workspace_entry_point(mode):
    from dlrm_amd import _capi, ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        _case(8192, 512, 256, 1)       # 64 row ranges of partial sums: most of the array
        _case(1000, 96, 64, 2)         # smaller layer right behind it
        _case(3000, 1, 256, 3)         # one output column: the streaming kernels' partial sums
        _case(2048, 512, 13, 4)        # 13 inputs
    finally:
        ops.set_gemm_mode(old)
    need = ops.linear_backward_weight_workspace_bytes(512, 4096, 4096)
    assert need > (40 << 20)
    dz = torch.zeros((512, 4096), device="cuda")
    x = torch.zeros((512, 4096), device="cuda")
    dw = torch.zeros((4096, 4096), device="cuda")
    with pytest.raises(_capi.DlrmError, match="static scratch"):
        ops.linear_backward_weight_static(dz, x, dw)


@pytest.mark.parametrize("mode", [1, 2], ids=["bf16x3", "f16x2"])
def test_static_scratch_entry_point_allocates_nothing(mode):
    """What the entry point is for: (1) the device's free memory is the same before and after a call, and (2) the
    call can be captured into a HIP graph -- where hipMalloc / hipFree are errors (capture mode "global") -- once a
    first call outside the capture has looked the array's address up; the replayed graph gives the workspace entry
    point's result bit for bit.  f16x2 included: its operand maxima come from a ring inside the code object and a
    captured memset."""
    from dlrm_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        M, N, K = 4096, 256, 384
        g = torch.Generator(device="cuda").manual_seed(11)
        dz = torch.randn((M, N), generator=g, device="cuda")
        x = torch.randn((M, K), generator=g, device="cuda")
        dw_ref, db_ref, _ = ops.linear_backward_weight(dz, x)
        dw = torch.full((N, K), float("nan"), device="cuda")
        db = torch.full((N,), float("nan"), device="cuda")
        ops.linear_backward_weight_static(dz, x, dw, db)          # first call on this device: address lookup
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            ops.linear_backward_weight_static(dz, x, dw, db)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        dw.fill_(float("nan"))
        db.fill_(float("nan"))
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="global"):
            ops.linear_backward_weight_static(dz, x, dw, db)
        torch.cuda.synchronize()
        assert torch.isnan(dw).all()                              # captured, not run
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        dw.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dw, dw_ref)
    finally:
        ops.set_gemm_mode(old)
