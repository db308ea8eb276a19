"""The reference's synthetic dataset, regenerated with the same generator calls.

MyDataset (src/dlrmtest.cpp:306-317): torch::manual_seed(0);
randn({N*B, dense}); randint(0, rows, {N*B, T*L}); randn({N*B, 1}); only the
first N rows are ever used (size() returns data_size, :328-331).  The CPU
generator stream of the container's torch reproduces the reference's ids: the
page-read totals of the reference's own CSVs depend on every id and match.
"""
import numpy as np
import torch


def reference_dataset(T, L, B, N, rows, dense=256):
    torch.manual_seed(0)
    x = torch.randn(N * B, dense)
    ids = torch.randint(0, rows, (N * B, T * L))
    y = torch.randn(N * B, 1)
    return x[:N].numpy().copy(), ids[:N].numpy().astype(np.int64), y[:N].numpy().copy()


def reference_batches(N, B):
    """(start, size) of every batch of one epoch; the last one may be short
    (src/dlrmtest.cpp:479-482)."""
    out = []
    s = 0
    while s < N:
        out.append((s, min(B, N - s)))
        s += B
    return out
