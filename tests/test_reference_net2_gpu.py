"""The HIP path held to the reference's own network run by the reference's own code: the same fixture and the same
walk as tests/test_oracle_reference_net2.py (which describes both), through the C-ABI entry points instead of the
oracle, in every evaluation of the float32 products."""
import numpy as np
import pytest
import torch

from tests.test_oracle_reference_net2 import DENSE, bags, close, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["f32mfma", "bf16x3", "f16x2"])
def test_hip_path_gives_the_reference_step(mode):
    from dlrm_amd import _capi, ops
    g = {k: torch.from_numpy(v).cuda() for k, v in load().items()}
    B = g["data"].shape[0]
    idx, off = (torch.from_numpy(a).cuda() for a in bags(B))
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        x = g["data"][:, :DENSE].contiguous()
        h1 = ops.linear_forward(x, g["fc1_weight"], g["fc1_bias"], True)
        z = torch.zeros((B, 96), device="cuda")
        ops.linear_forward(h1, g["fc2_weight"], g["fc2_bias"], True, out=z[:, :32])
        for t, name in enumerate(["emb0", "emb1"]):
            ops.embbag_forward(g[name + "_weight"], idx, off, B, pooling=_capi.POOL_MEAN, include_last_offset=True,
                               out=z[:, 32 + 32 * t:], out_ld=96)
        h3 = ops.linear_forward(z, g["fc3_weight"], g["fc3_bias"], True)
        pred = ops.linear_forward(h3, g["fc4_weight"], g["fc4_bias"], True)
        loss, dpred = ops.loss_mse(pred, g["target"], relu_mask_pred=True)
        d3 = ops.linear_backward_data(dpred, g["fc4_weight"], relu_mask=h3)
        dw4, db4, _ = ops.linear_backward_weight(dpred, h3)
        dz = ops.linear_backward_data(d3, g["fc3_weight"])
        dw3, db3, _ = ops.linear_backward_weight(d3, z)
        d2 = (dz[:, :32] * (z[:, :32] > 0)).contiguous()
        d1 = ops.linear_backward_data(d2, g["fc2_weight"], relu_mask=h1)
        dw2, db2, _ = ops.linear_backward_weight(d2, h1)
        dw1, db1, _ = ops.linear_backward_weight(d1, x)
        moved = {}
        for t, name in enumerate(["emb0", "emb1"]):
            w = g[name + "_weight"].clone()
            ops.embbag_backward_update(w, idx, off, B, dz[:, 32 + 32 * t:64 + 32 * t], pooling=_capi.POOL_MEAN,
                                       include_last_offset=True, lr=1.0, grad_ld=96)
            moved[name] = g[name + "_weight"] - w
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_mode(old)
    ref = {k: v.cpu().numpy() for k, v in g.items()}
    assert close(pred.cpu().numpy(), ref["prediction"])
    assert abs(float(loss) - float(ref["loss"][0])) <= 2e-5 * float(ref["loss"][0])
    for name, (dw, db) in [("fc4", (dw4, db4)), ("fc3", (dw3, db3)), ("fc2", (dw2, db2)), ("fc1", (dw1, db1))]:
        assert close(dw.cpu().numpy(), ref[name + "_weight_grad"]), name
        assert close(db.cpu().numpy(), ref[name + "_bias_grad"]), name
    for name in moved:
        assert close(moved[name].cpu().numpy(), ref[name + "_weight_grad"], 5e-5), name
