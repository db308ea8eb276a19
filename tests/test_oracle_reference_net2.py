"""The CPU oracle held to the reference's own network, run by the reference's own code.

tests/golden/reference_net2_step.json is what oracle/ref/net2_harness.cpp printed: the reference's toy driver
(dlrm_libtorch/dlrmtest.cpp, its struct Net2 at :129-240 and the loop body at :339-372) compiled from its own source
against libtorch, run on one batch, with every tensor dumped -- the weights it bound, the two EmbeddingBag tables
its create_emb (:20-52) built, its prediction, its mse_loss and every gradient of loss.backward().  `make -C oracle
ref-golden` regenerates the file where the reference tree is present.

The path, as the reference executes it: both bags look up rows 0..7 as two bags of four (:172-191, the ids in the batch
are read and not used), pooled by the mean (from_pretrained is a static factory: the options of the module it is
called on, kSum among them, are dropped); relu(fc1), relu(fc2) on the 256 dense features; [dense | bag0 | bag1];
relu(fc3), relu(fc4); mse_loss with the mean over the batch."""
import json
import os

import numpy as np
import pytest

PATH = os.path.join(os.path.dirname(__file__), "golden", "reference_net2_step.json")
DENSE, LOOKUPS = 256, 4


def load():
    with open(PATH) as f:
        raw = json.load(f)
    return {k: np.asarray(v["data"], np.float32).reshape(v["shape"]) for k, v in raw.items() if isinstance(v, dict)}


def close(a, ref, rel=2e-5):
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    return float(np.max(np.abs(np.asarray(a, np.float64) - ref))) <= rel * scale


def bags(batch):
    """the ids Net2::forward builds (dlrmtest.cpp:172-191): 0 .. batch*4-1, four per sample"""
    return np.arange(batch * LOOKUPS, dtype=np.int64), np.arange(0, batch * LOOKUPS + 1, LOOKUPS, dtype=np.int64)


def test_fixture_is_what_the_reference_harness_prints():
    """where the harness has been built here (the reference tree is present), it reproduces the committed file"""
    exe = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "net2_harness")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/net2_harness is not built (needs the reference tree: make -C oracle ref)")
    import subprocess
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    with open(PATH) as f:
        assert json.loads(out.stdout) == json.load(f)


def test_oracle_forward_is_the_reference_network(oracle):
    g = load()
    B = g["data"].shape[0]
    idx, off = bags(B)
    h = oracle.linear_forward(g["data"][:, :DENSE], g["fc1_weight"], g["fc1_bias"], True)
    h = oracle.linear_forward(h, g["fc2_weight"], g["fc2_bias"], True)
    e0 = oracle.embbag_forward(g["emb0_weight"], idx, off, B, pooling=oracle.POOL_MEAN, include_last_offset=True)
    e1 = oracle.embbag_forward(g["emb1_weight"], idx, off, B, pooling=oracle.POOL_MEAN, include_last_offset=True)
    z = np.concatenate([h, e0, e1], 1)
    h3 = oracle.linear_forward(z, g["fc3_weight"], g["fc3_bias"], True)
    pred = oracle.linear_forward(h3, g["fc4_weight"], g["fc4_bias"], True)
    assert close(pred, g["prediction"])
    loss, _ = oracle.mse_loss(pred, g["target"])
    assert abs(loss - float(g["loss"][0])) <= 2e-5 * abs(float(g["loss"][0]))
    # the sum pooling the module options name would not give the reference's prediction: the mean is what runs
    s0 = oracle.embbag_forward(g["emb0_weight"], idx, off, B, pooling=oracle.POOL_SUM, include_last_offset=True)
    assert not close(s0, e0)


def test_oracle_backward_is_the_reference_backward(oracle):
    g = load()
    B = g["data"].shape[0]
    idx, off = bags(B)
    x = g["data"][:, :DENSE]
    h1 = oracle.linear_forward(x, g["fc1_weight"], g["fc1_bias"], True)
    h2 = oracle.linear_forward(h1, g["fc2_weight"], g["fc2_bias"], True)
    e0 = oracle.embbag_forward(g["emb0_weight"], idx, off, B, pooling=oracle.POOL_MEAN, include_last_offset=True)
    e1 = oracle.embbag_forward(g["emb1_weight"], idx, off, B, pooling=oracle.POOL_MEAN, include_last_offset=True)
    z = np.concatenate([h2, e0, e1], 1)
    h3 = oracle.linear_forward(z, g["fc3_weight"], g["fc3_bias"], True)
    pred = oracle.linear_forward(h3, g["fc4_weight"], g["fc4_bias"], True)
    _, dpred = oracle.mse_loss(pred, g["target"])
    d3, dw4, db4 = oracle.linear_backward(h3, g["fc4_weight"], pred, dpred.reshape(B, 1), True)
    dz, dw3, db3 = oracle.linear_backward(z, g["fc3_weight"], h3, d3, True)
    d1, dw2, db2 = oracle.linear_backward(h1, g["fc2_weight"], h2, np.ascontiguousarray(dz[:, :32]), True)
    _, dw1, db1 = oracle.linear_backward(x, g["fc1_weight"], h1, d1, True, need_dx=False)
    for name, got in [("fc4", (dw4, db4)), ("fc3", (dw3, db3)), ("fc2", (dw2, db2)), ("fc1", (dw1, db1))]:
        assert close(got[0], g[name + "_weight_grad"]), name
        assert close(got[1], g[name + "_bias_grad"]), name
    # the bag gradient: one SGD step of rate 1 moves each row by minus its gradient
    for t, name in enumerate(["emb0", "emb1"]):
        w = g[name + "_weight"].copy()
        grad = np.ascontiguousarray(dz[:, 32 + 32 * t:64 + 32 * t])
        oracle.embbag_backward_update(w, idx, off, B, grad, pooling=oracle.POOL_MEAN, include_last_offset=True, lr=1.0)
        assert close(g[name + "_weight"] - w, g[name + "_weight_grad"], 5e-5), name
