"""Generates golden vectors for the floating point operators of the hot path
with libtorch's own CPU kernels (python torch dispatches to the same ATen
operators the reference calls from C++: torch::embedding_bag
src/dlrmtest.cpp:191, torch::nn::Linear :260-270, torch::relu, torch::cat :265,
torch::mse_loss :522, backward :531, torch::optim::SGD :505-533).

The reference itself has no test or fixture for these values, so this is the
pin for the oracle's restatement: run in the build container,

    python tests/golden/make_golden.py

writes tests/golden/ops_golden.npz (inputs and expected outputs only).
"""
import os

import numpy as np
import torch

F = torch.nn.functional


def embbag_case(rng, T, B, D, rows, L, mode, ragged):
    g = {}
    n_bags = T * B
    lens = rng.integers(0, 2 * L + 1, size=n_bags) if ragged else np.full(n_bags, L)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tro = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    idx = np.empty(int(off[-1]), dtype=np.int64)
    for t in range(T):
        s, e = off[t * B], off[(t + 1) * B]
        idx[s:e] = rng.integers(0, rows[t], size=e - s)
    w = rng.standard_normal((int(tro[-1]), D)).astype(np.float32)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    outs, new_w = [], []
    for t in range(T):
        wt = torch.from_numpy(w[tro[t]:tro[t + 1]].copy()).requires_grad_(True)
        s, e = off[t * B], off[(t + 1) * B]
        o = F.embedding_bag(torch.from_numpy(idx[s:e]), wt, torch.from_numpy(off[t * B:(t + 1) * B] - s), mode=mode)
        outs.append(o.detach().numpy())
        o.backward(torch.from_numpy(grad[:, t * D:(t + 1) * D].copy()))
        opt = torch.optim.SGD([wt], lr=0.05)  # plain SGD == the reference's fresh-momentum step
        opt.step()
        new_w.append(wt.detach().numpy())
    g.update(w=w, tro=tro, idx=idx, off=off, grad=grad, out=np.concatenate(outs, axis=1),
             w_sgd=np.concatenate(new_w, axis=0), B=np.int64(B), T=np.int64(T), mean=np.int64(mode == "mean"))
    return g


def mlp_case(rng, M, sizes, final_relu):
    x = torch.from_numpy(rng.standard_normal((M, sizes[0])).astype(np.float32))
    Ws = [torch.from_numpy((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)).requires_grad_(True)
          for k, n in zip(sizes[:-1], sizes[1:])]
    bs = [torch.from_numpy(rng.standard_normal(n).astype(np.float32) * 0.1).requires_grad_(True) for n in sizes[1:]]
    xin = x.clone().requires_grad_(True)
    y = xin
    for i, (W, b) in enumerate(zip(Ws, bs)):
        y = F.linear(y, W, b)
        if final_relu or i + 1 < len(Ws):
            y = torch.relu(y)
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    g = {"x": x.numpy(), "y": y.detach().numpy(), "dy": dy.numpy(), "dx": xin.grad.numpy(),
         "n_layers": np.int64(len(Ws)), "final_relu": np.int64(final_relu)}
    for i, (W, b) in enumerate(zip(Ws, bs)):
        g["W%d" % i], g["b%d" % i] = W.detach().numpy(), b.detach().numpy()
        g["dW%d" % i], g["db%d" % i] = W.grad.numpy(), b.grad.numpy()
    return g


def interact_case(rng, B, Fe, D):
    x = torch.from_numpy(rng.standard_normal((B, Fe, D)).astype(np.float32)).requires_grad_(True)
    z = torch.bmm(x, x.transpose(1, 2))
    li, lj = torch.tril_indices(Fe, Fe, offset=-1)
    out = torch.cat([x[:, 0, :], z[:, li, lj]], dim=1)
    dout = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
    out.backward(dout)
    return {"x": x.detach().numpy(), "out": out.detach().numpy(), "dout": dout.numpy(), "dx": x.grad.numpy()}


def loss_case(rng, n):
    p = torch.from_numpy((rng.standard_normal(n) * 2).astype(np.float32)).requires_grad_(True)
    t = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    lab = torch.from_numpy((rng.random(n) < 0.3).astype(np.float32))
    l1 = F.mse_loss(p, t)
    l1.backward()
    d1 = p.grad.numpy().copy()
    p.grad = None
    l2 = F.binary_cross_entropy_with_logits(p, lab)
    l2.backward()
    return {"pred": p.detach().numpy(), "target": t.numpy(), "label": lab.numpy(), "mse": np.float32(l1.item()),
            "dmse": d1, "bce": np.float32(l2.item()), "dbce": p.grad.numpy()}


def as_shipped_step(rng):
    """One batch of the reference loop exactly as it executes
    (src/dlrmtest.cpp:53-107, :170-206, :499-535): per table a mini-table of the batch's SORTED ids, mean pooling
    over consecutive groups of L sorted ids, concat with the bottom MLP output,
    ReLU on every layer, MSE; SGD(lr 0.01) updates the mini-table of table 0
    only; rows are written back in sorted order (last duplicate wins)."""
    T, D, L, B, dense, rows = 3, 32, 4, 5, 16, 40
    tables = [rng.random((rows, D)).astype(np.float32) for _ in range(T)]
    ids = rng.integers(0, rows, size=(B, T * L))
    x = rng.standard_normal((B, dense)).astype(np.float32)
    target = rng.standard_normal((B, 1)).astype(np.float32)
    sizes_b, top_in = [dense, 24, 8], 8 + T * D
    W = [rng.standard_normal((24, dense)).astype(np.float32), rng.standard_normal((8, 24)).astype(np.float32),
         rng.standard_normal((8, top_in)).astype(np.float32) * 0.1, rng.standard_normal((1, 8)).astype(np.float32)]
    bias = [np.abs(rng.standard_normal(n)).astype(np.float32) for n in (24, 8, 8, 1)]
    minis, sorted_ids = [], []
    for t in range(T):
        s = np.sort(ids[:, t * L:(t + 1) * L].reshape(-1))
        sorted_ids.append(s)
        minis.append(torch.from_numpy(tables[t][s].copy()).requires_grad_(True))
    lookup = torch.arange(B * L).view(B, L)
    pooled = [F.embedding_bag(lookup, m, mode="mean") for m in minis]
    h = torch.from_numpy(x)
    h = torch.relu(F.linear(h, torch.from_numpy(W[0]), torch.from_numpy(bias[0])))
    h = torch.relu(F.linear(h, torch.from_numpy(W[1]), torch.from_numpy(bias[1])))
    h = torch.cat([h] + pooled, dim=1)
    h = torch.relu(F.linear(h, torch.from_numpy(W[2]), torch.from_numpy(bias[2])))
    pred = torch.relu(F.linear(h, torch.from_numpy(W[3]), torch.from_numpy(bias[3])))
    loss = F.mse_loss(pred, torch.from_numpy(target))
    loss.backward()
    new_tables = [t.copy() for t in tables]
    upd = (minis[0] - 0.01 * minis[0].grad).detach().numpy()
    for y, rid in enumerate(sorted_ids[0]):
        new_tables[0][rid] = upd[y]
    g = {"ids": ids.astype(np.int64), "x": x, "target": target, "pred": pred.detach().numpy(),
         "loss": np.float32(loss.item()), "T": np.int64(T), "L": np.int64(L)}
    for t in range(T):
        g["table%d" % t], g["new_table%d" % t] = tables[t], new_tables[t]
    for i in range(4):
        g["W%d" % i], g["b%d" % i] = W[i], bias[i]
    return g


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    rng = np.random.default_rng(20240607)
    out = {}

    def put(prefix, d):
        for k, v in d.items():
            out[prefix + "/" + k] = v

    put("emb_sum", embbag_case(rng, 3, 9, 16, [20, 3, 50], 2, "sum", False))
    put("emb_mean_ragged", embbag_case(rng, 2, 7, 32, [11, 40], 3, "mean", True))
    put("emb_rmc1_bag", embbag_case(rng, 1, 4, 32, [200], 80, "mean", False))
    put("mlp_ref", mlp_case(rng, 6, [256, 128, 32], True))
    put("mlp_top", mlp_case(rng, 9, [47, 64, 16, 1], False))
    put("interact", interact_case(rng, 4, 27, 16))
    put("interact_small", interact_case(rng, 3, 3, 8))
    put("loss", loss_case(rng, 37))
    put("as_shipped", as_shipped_step(rng))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
