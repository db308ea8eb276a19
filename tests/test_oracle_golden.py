"""Oracle vs golden vectors produced by libtorch's CPU kernels
(tests/golden/make_golden.py -> tests/golden/ops_golden.npz).  Pins the
oracle's restatement of torch::embedding_bag, Linear/ReLU, the dot
interaction, the losses, SGD, and the reference loop as it executes."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops_golden.npz"), allow_pickle=False)


def case(prefix):
    return {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(prefix + "/")}


def close(a, ref, rel=1e-5):
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    return float(np.max(np.abs(np.asarray(a) - ref))) <= rel * scale


@pytest.mark.parametrize("name", ["emb_sum", "emb_mean_ragged", "emb_rmc1_bag"])
def test_embedding_bag_forward_and_sgd(oracle, name):
    c = case(name)
    B, T = int(c["B"]), int(c["T"])
    pooling = oracle.POOL_MEAN if int(c["mean"]) else oracle.POOL_SUM
    out = oracle.embbag_forward(c["w"], c["idx"], c["off"], B, T, c["tro"], pooling, include_last_offset=True)
    assert close(out, c["out"])
    # without the trailing offset as well
    out2 = oracle.embbag_forward(c["w"], c["idx"], c["off"][:-1], B, T, c["tro"], pooling, include_last_offset=False)
    assert np.array_equal(out, out2)
    w = c["w"].copy()
    oracle.embbag_backward_update(w, c["idx"], c["off"], B, c["grad"], T, c["tro"], pooling, True, lr=0.05)
    assert close(w, c["w_sgd"])


@pytest.mark.parametrize("name", ["mlp_ref", "mlp_top"])
def test_mlp_forward_backward(oracle, name):
    c = case(name)
    n, final_relu = int(c["n_layers"]), bool(int(c["final_relu"]))
    acts = [c["x"]]
    for i in range(n):
        acts.append(oracle.linear_forward(acts[-1], c["W%d" % i], c["b%d" % i], final_relu or i + 1 < n))
    assert close(acts[-1], c["y"])
    dy = c["dy"]
    for i in reversed(range(n)):
        dx, dw, db = oracle.linear_backward(acts[i], c["W%d" % i], acts[i + 1], dy, final_relu or i + 1 < n)
        assert close(dw, c["dW%d" % i]) and close(db, c["db%d" % i])
        dy = dx
    assert close(dy, c["dx"])


@pytest.mark.parametrize("name", ["interact", "interact_small"])
def test_dot_interaction(oracle, name):
    c = case(name)
    assert close(oracle.interact_dot_forward(c["x"]), c["out"])
    assert close(oracle.interact_dot_backward(c["x"], c["dout"]), c["dx"])


def test_losses(oracle):
    c = case("loss")
    l, d = oracle.mse_loss(c["pred"], c["target"])
    assert abs(l - float(c["mse"])) <= 1e-6 * abs(float(c["mse"])) and close(d, c["dmse"])
    l, d = oracle.bce_logits_loss(c["pred"], c["label"])
    assert abs(l - float(c["bce"])) <= 1e-6 * abs(float(c["bce"])) and close(d, c["dbce"])


def as_shipped_step_with(orc, c):
    from tests.oracle_model import as_shipped_step
    T, L = int(c["T"]), int(c["L"])
    tables = [c["table%d" % t].copy() for t in range(T)]
    loss, pred = as_shipped_step(orc, tables, c["ids"], c["x"], c["target"], [c["W%d" % i] for i in range(4)],
                                 [c["b%d" % i] for i in range(4)], L)
    return loss, pred, tables


def test_reference_loop_as_shipped(oracle):
    c = case("as_shipped")
    loss, pred, tables = as_shipped_step_with(oracle, c)
    assert abs(loss - float(c["loss"])) <= 1e-5 * abs(float(c["loss"]))
    assert close(pred, c["pred"])
    for t in range(int(c["T"])):
        assert close(tables[t], c["new_table%d" % t])
    # tables other than 0 are written back unchanged (only table 0 is in the optimizer)
    assert np.array_equal(tables[1], c["table1"])
