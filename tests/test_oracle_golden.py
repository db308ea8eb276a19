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


def test_rowwise_adagrad_against_libtorch_adagrad(oracle):
    """libtorch has no row-wise Adagrad, but where every element of a row's gradient has the same magnitude the
    row mean of g^2 IS each element's g^2, and the row-wise rule (state += mean g^2; w -= lr g / (sqrt(state) +
    eps)) must give what torch.optim.Adagrad gives element by element.  Pins the rule's arithmetic (accumulate,
    then sqrt, eps outside the root) against libtorch's optimizer over three steps."""
    import torch
    rng = np.random.default_rng(8)
    rows, D, B, lr, eps = 50, 16, 20, 0.05, 1e-8
    w0 = rng.standard_normal((rows, D)).astype(np.float32)
    w = w0.copy()
    state = np.zeros(rows, np.float32)
    p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    opt = torch.optim.Adagrad([p], lr=lr, eps=eps, initial_accumulator_value=0.0, lr_decay=0.0)
    for step in range(3):
        idx = rng.permutation(rows)[:B].astype(np.int64)                  # distinct rows: one bag each
        off = np.arange(B + 1, dtype=np.int64)
        mag = rng.uniform(0.1, 2.0, size=(B, 1)).astype(np.float32)
        g = (mag * rng.choice([-1.0, 1.0], size=(B, D))).astype(np.float32)
        oracle.embbag_backward_update(w, idx, off, B, g, include_last_offset=True, optimizer=oracle.OPT_ADAGRAD_ROWWISE,
                                      lr=lr, eps=eps, adagrad_state=state)
        dense = torch.zeros(rows, D)
        dense[torch.from_numpy(idx)] = torch.from_numpy(g)
        # torch's rule touches every element; rows outside the batch have g = 0: state and value unchanged
        opt.zero_grad()
        p.grad = dense
        opt.step()
        assert np.allclose(w, p.detach().numpy(), rtol=2e-6, atol=1e-7), "step %d" % step
    assert np.allclose(state, opt.state[p]["sum"].numpy()[:, 0], rtol=1e-6)
