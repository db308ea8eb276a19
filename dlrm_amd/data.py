"""Synthetic batches (the reference's MyDataset, src/dlrmtest.cpp:299-334,
generates randn dense features, uniform randint ids and randn labels; ids are
produced here as integers, not float-encoded).  Generation uses torch's device
RNG: it is input plumbing, not part of the measured hot path."""
import torch


def zipf_ids(n, rows, alpha, gen, device):
    """Approximate Zipf(alpha) ranks in [0, rows) by inverting the continuous
    power-law CDF, then scatter ranks over the table with a multiplicative hash
    so that hot rows are not neighbours."""
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    if abs(alpha - 1.0) < 1e-9:
        r = torch.exp(u * torch.log(torch.tensor(float(rows), dtype=torch.float64)))
    else:
        a = 1.0 - alpha
        r = ((float(rows) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    r = (r.floor().to(torch.int64) - 1).clamp_(0, rows - 1)
    return (r * 2654435761 + 12345) % rows


def synthetic_batch(table_rows, batch, lookups, dense_in, device, seed, loss="bce", dist="uniform", alpha=1.05,
                    idx_dtype=torch.int32, local_batch=None, shared_from=None, shared_seed=0):
    """Returns (dense [local_batch, dense_in], indices [T*batch*lookups], offsets [T*batch+1], labels).
    Bags are table-major (g = t*batch + b), fixed `lookups` rows per bag.  Tables shared_from..
    draw their ids from a generator seeded with `shared_seed` only: row-wise split tables must see
    the same global batch on every rank."""
    gen = torch.Generator(device=device).manual_seed(seed)
    gen_shared = torch.Generator(device=device).manual_seed(shared_seed)
    T = len(table_rows)
    lb = batch if local_batch is None else local_batch
    dense = torch.randn((lb, dense_in), generator=gen, device=device)
    if loss == "bce":
        labels = (torch.rand(lb, generator=gen, device=device) < 0.25).float()
    else:
        labels = torch.randn(lb, generator=gen, device=device)
    n_per = batch * lookups
    idx = torch.empty(T * n_per, dtype=idx_dtype, device=device)
    for t, rows in enumerate(table_rows):
        g = gen_shared if shared_from is not None and t >= shared_from else gen
        if dist == "zipf":
            v = zipf_ids(n_per, rows, alpha, g, device)
        else:
            v = torch.randint(0, rows, (n_per,), generator=g, device=device, dtype=torch.int64)
        idx[t * n_per:(t + 1) * n_per] = v.to(idx_dtype)
    off = (torch.arange(T * batch + 1, device=device, dtype=torch.int64) * lookups).to(idx_dtype)
    return dense, idx, off, labels
