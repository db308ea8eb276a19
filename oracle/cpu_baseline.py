"""CPU baseline: the reference's operator sequence on libtorch CPU kernels.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (used by bench.py's cpu_baseline leg and
by tests as a second checker).  The reference calls torch::embedding_bag,
torch::nn::Linear, torch::relu, torch::cat, a loss, loss.backward() and
torch::optim::SGD::step from C++ (src/dlrmtest.cpp:191-197, :254-272,
:505-533); python torch dispatches to the same ATen CPU kernels, so this file
issues that sequence (plus the bmm + lower-triangle dot interaction and BCE
loss the Criteo shapes need) with every host thread ATen is given.
"""
import time

import torch


def build_cpu_model(cfg, table_rows_cap, seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = [min(r, table_rows_cap) for r in cfg.table_rows]
    tables = [torch.rand((r, cfg.dim), generator=g).requires_grad_(True) for r in rows]

    def mlp(sizes):
        ps = []
        for K, N in zip(sizes[:-1], sizes[1:]):
            ps.append((torch.randn(N, K, generator=g) * (2.0 / (K + N)) ** 0.5).requires_grad_(True))
            ps.append((torch.randn(N, generator=g) * (1.0 / N) ** 0.5).requires_grad_(True))
        return ps

    bottom = mlp([cfg.dense_in] + list(cfg.bottom))
    top = mlp([cfg.top_in] + list(cfg.top))
    return rows, tables, bottom, top


def cpu_train_step(cfg, tables, bottom, top, dense, idx_list, off, labels, lr):
    """idx_list[t]: int64 [B*L]; off: int64 [B] bag starts."""
    F = torch.nn.functional
    x = dense
    for i in range(0, len(bottom), 2):
        x = torch.relu(F.linear(x, bottom[i], bottom[i + 1]))
    mode = "mean" if cfg.pooling == "mean" else "sum"
    pooled = [F.embedding_bag(idx_list[t], tables[t], off, mode=mode, sparse=True) for t in range(len(tables))]
    if cfg.interaction == "dot":
        B = dense.shape[0]
        feat = torch.cat([x] + pooled, dim=1).view(B, len(tables) + 1, cfg.dim)
        z = torch.bmm(feat, feat.transpose(1, 2))
        li, lj = torch.tril_indices(feat.shape[1], feat.shape[1], offset=-1)
        inter = torch.cat([x, z[:, li, lj]], dim=1)
    else:
        inter = torch.cat([x] + pooled, dim=1)
    y = inter
    n_top = len(top) // 2
    for i in range(n_top):
        y = F.linear(y, top[2 * i], top[2 * i + 1])
        if cfg.final_relu or i + 1 < n_top:
            y = torch.relu(y)
    y = y.view(-1)
    if cfg.loss == "bce":
        loss = F.binary_cross_entropy_with_logits(y, labels)
    else:
        loss = F.mse_loss(y, labels)
    loss.backward()
    with torch.no_grad():
        for p in bottom + top:
            p -= lr * p.grad
            p.grad = None
        for w in tables:
            g = w.grad.coalesce()  # duplicates accumulate, one update per touched row
            w.index_add_(0, g.indices()[0], g.values(), alpha=-lr)
            w.grad = None
    return float(loss.detach())


def _time_steps(cfg, tables, bottom, top, rows, batch, warmup, steps, seed):
    g = torch.Generator().manual_seed(seed + 1)
    dense = torch.randn(batch, cfg.dense_in, generator=g)
    labels = (torch.rand(batch, generator=g) < 0.25).float() if cfg.loss == "bce" else torch.randn(batch, generator=g)
    L = cfg.lookups
    idx_list = [torch.randint(0, r, (batch * L,), generator=g) for r in rows]
    off = torch.arange(batch, dtype=torch.int64) * L
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        cpu_train_step(cfg, tables, bottom, top, dense, idx_list, off, labels, cfg.lr)
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
    times.sort()
    return times[len(times) // 2]


def time_cpu_baseline(cfg, batch=65536, table_rows_cap=2_000_000, warmup=2, steps=5, one_thread_batch=8192, seed=0):
    """Median step time of `steps` train steps after `warmup` on every host thread ATen is given, at the
    workload's own batch, plus a one-thread run on a smaller batch (samples/s scale with neither).  Tables are
    capped at table_rows_cap rows: building 96 GB of tables on the host would take longer than the whole
    benchmark; the cap is reported.  Returns the cpu_baseline object of the bench line."""
    rows, tables, bottom, top = build_cpu_model(cfg, table_rows_cap, seed)
    threads = torch.get_num_threads()
    med = _time_steps(cfg, tables, bottom, top, rows, batch, warmup, steps, seed)
    torch.set_num_threads(1)
    try:
        b1 = min(one_thread_batch, batch)
        med1 = _time_steps(cfg, tables, bottom, top, rows, b1, 1, 3, seed)
    finally:
        torch.set_num_threads(threads)
    return {
        "value": batch / med,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "table_rows_cap": table_rows_cap,
        "table_GB": sum(rows) * cfg.dim * 4 / 1e9,
        "one_thread": {"value": b1 / med1, "unit": "samples/s", "cores": 1, "batch": b1, "median_step_s": med1},
        "sample": "%d train steps (after %d warm-ups) of batch %d, same model shape, tables capped at %d rows "
                  "(%.1f GB on the host), libtorch CPU kernels via ATen on %d threads, median step %.3f s; "
                  "one thread: 3 steps of batch %d, median %.3f s"
                  % (steps, warmup, batch, table_rows_cap, sum(rows) * cfg.dim * 4 / 1e9, threads, med, b1, med1),
    }
