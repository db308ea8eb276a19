#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP-event timed, inputs resident in HBM).

    python tools/microbench.py gemm            # MLP GEMM shapes of the MLPerf model
    python tools/microbench.py embbag          # BASELINE config 5: L x D x mode x dtype sweep
    python tools/microbench.py sparse          # fused sparse backward, uniform and Zipf ids
    python tools/microbench.py interact
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import _capi, ops  # noqa: E402
from dlrm_amd.data import zipf_ids  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def pad4(n):
    return (n + 3) // 4 * 4


def gemm(shapes=None):
    M = int(os.environ.get("GEMM_M", "65536"))
    shapes = shapes or [(1024, 1024), (1024, 479), (512, 1024), (256, 512), (512, 13), (128, 256), (1, 256)]
    pad = (lambda n: n if n <= 128 else (n + 127) // 128 * 128) if os.environ.get("GEMM_PADK") else pad4
    out = []
    for N, K in shapes:
        x = torch.randn(M, pad(pad4(K)), device="cuda")[:, :K]
        w = torch.randn(N, pad(pad4(K)), device="cuda")[:, :K]
        b = torch.randn(N, device="cuda")
        if os.environ.get("GEMM_ZERO"):      # operands of zeros: how much of the time is data-dependent power?
            x.zero_(); w.zero_()
        y = torch.empty(M, pad(pad4(N)), device="cuda")[:, :N]
        dz = torch.randn(M, pad(pad4(N)), device="cuda")[:, :N]
        if os.environ.get("GEMM_RAWBF16"):   # every 16-bit half of the operands is a sane random bf16 value
            for t in (x, w, dz):
                base = t._base if t._base is not None else t
                raw = torch.randn(base.shape[0], base.shape[1] * 2, device="cuda").to(torch.bfloat16)
                base.copy_(raw.view(torch.float32))
        dx = torch.empty(M, pad(pad4(K)), device="cuda")[:, :K]
        dw = torch.empty(N, pad(pad4(K)), device="cuda")[:, :K]
        db = torch.empty(N, device="cuda")
        ws = torch.empty(ops.linear_backward_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: ops.linear_forward(x, w, b, True, out=y))
        t_d = timeit(lambda: ops.linear_backward_data(dz, w, relu_mask=x, out=dx))
        t_w = timeit(lambda: ops.linear_backward_weight(dz, x, dw=dw, db=db, workspace=ws))
        rec = {"M": M, "N": N, "K": K, "fwd_TF": fl / t_f / 1e12, "bwd_data_TF": fl / t_d / 1e12,
               "bwd_weight_TF": fl / t_w / 1e12, "fwd_us": t_f * 1e6, "bwd_data_us": t_d * 1e6, "bwd_weight_us": t_w * 1e6}
        out.append(rec)
        print(json.dumps(rec))
    return out


def embbag(rows=10_000_000, B=65536):
    out = []
    for D in (64, 128, 256):
        for dt in (torch.float32, torch.bfloat16):
            w = torch.empty(rows, D, device="cuda", dtype=dt)
            ops.table_init_uniform_(w, 1)
            for L in (1, 2, 4, 8, 16, 32, 64, 128):
                n = B * L
                for dist in ("uniform", "zipf"):
                    g = torch.Generator(device="cuda").manual_seed(L)
                    if dist == "zipf":
                        idx = zipf_ids(n, rows, 1.05, g, "cuda").to(torch.int32)
                    else:
                        idx = torch.randint(0, rows, (n,), generator=g, device="cuda", dtype=torch.int32)
                    off = (torch.arange(B + 1, device="cuda") * L).to(torch.int32)
                    o = torch.empty(B, D, device="cuda")
                    for mode in ("sum", "mean"):
                        pool = _capi.POOL_MEAN if mode == "mean" else _capi.POOL_SUM
                        t = timeit(lambda: ops.embbag_forward(w, idx, off, B, 1, None, pool, True, out=o, out_ld=D),
                                   iters=5, warmup=1)
                        byt = n * D * w.element_size() + B * D * 4 + n * 4 + (B + 1) * 4
                        rec = {"op": "fwd", "D": D, "rows": "bf16" if dt == torch.bfloat16 else "f32", "L": L,
                               "ids": dist, "mode": mode, "us": t * 1e6, "GBps": byt / t / 1e9}
                        out.append(rec)
                        print(json.dumps(rec))
            del w
    return out


def sparse(rows=10_000_000, B=65536):
    """Fused sparse update, whole call (expand + sort + apply + combine): row width x row type x ids per bag x
    id distribution x (sum|mean pooling) x (SGD | row-wise Adagrad).  Adagrad's bytes add the state row read and
    written (one float per row for the row-wise form)."""
    out = []
    for D in (64, 128, 256):
        for dt in (torch.float32, torch.bfloat16):
            w = torch.empty(rows, D, device="cuda", dtype=dt)
            ops.table_init_uniform_(w, 1)
            state = torch.zeros(rows, device="cuda")
            for L in (1, 8, 32, 128):
                n = B * L
                for dist in ("uniform", "zipf"):
                    g = torch.Generator(device="cuda").manual_seed(L)
                    if dist == "zipf":
                        idx = zipf_ids(n, rows, 1.05, g, "cuda").to(torch.int32)
                    else:
                        idx = torch.randint(0, rows, (n,), generator=g, device="cuda", dtype=torch.int32)
                    U = int(torch.unique(idx).numel())
                    off = (torch.arange(B + 1, device="cuda") * L).to(torch.int32)
                    grad = torch.randn(B, D, device="cuda")
                    for mode, opt in (("sum", "sgd"), ("mean", "sgd"), ("sum", "adagrad")):
                        ws = [None]
                        pool = _capi.POOL_MEAN if mode == "mean" else _capi.POOL_SUM
                        kw = dict(optimizer=_capi.OPT_ADAGRAD_ROWWISE, opt_state=state) if opt == "adagrad" else {}

                        def run():
                            ws[0] = ops.embbag_backward_update(w, idx, off, B, grad, 1, None, pool, True,
                                                               lr=0.0, workspace=ws[0], **kw)
                        t = timeit(run, iters=5, warmup=1)
                        # the second half alone (what a train step has on its critical path once the sort of the
                        # ids runs beside the forward pass on another stream)
                        ws[0] = ops.embbag_backward_prepare(w, idx, off, B, 1, None, pool, True, workspace=ws[0])
                        t_a = timeit(lambda: ops.embbag_backward_apply(w, idx, off, B, grad, ws[0], 1, None, pool, True,
                                                                       lr=0.0, **kw), iters=5, warmup=1)
                        byt = B * D * 4 + 2 * U * D * w.element_size() + n * 4 + (2 * U * 4 if opt == "adagrad" else 0)
                        rec = {"op": "bwd", "opt": opt, "mode": mode, "D": D,
                               "rows": "bf16" if dt == torch.bfloat16 else "f32", "L": L,
                               "ids": dist, "unique": U, "us": t * 1e6, "GBps": byt / t / 1e9,
                               "apply_us": t_a * 1e6, "apply_GBps": byt / t_a / 1e9}
                        out.append(rec)
                        print(json.dumps(rec))
            del w, state
    return out


def sparse_mlperf(B=65536, D=128):
    """Fused sparse update on subsets of the MLPerf tables (one id per bag): where does the time go?"""
    from dlrm_amd.model import MLPERF_TABLE_ROWS as R
    groups = {"huge(>=20M)": [r for r in R if r >= 20_000_000], "mid(10k..20M)": [r for r in R if 10_000 <= r < 20_000_000],
              "small(200..10k)": [r for r in R if 200 <= r < 10_000], "tiny(<200)": [r for r in R if r < 200],
              "all": list(R)}
    out = []
    for name, rows in groups.items():
        T = len(rows)
        tro = torch.tensor([0] + list(torch.tensor(rows).cumsum(0)), dtype=torch.int64, device="cuda")
        w = torch.empty(int(tro[-1]), D, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(3)
        idx = torch.cat([torch.randint(0, r, (B,), generator=g, device="cuda", dtype=torch.int32) for r in rows])
        off = torch.arange(T * B + 1, device="cuda", dtype=torch.int32)
        grad = torch.randn(B, T * D, device="cuda")
        o = torch.empty(B, T * D, device="cuda")
        ws = [None]

        def run():
            ws[0] = ops.embbag_backward_update(w, idx, off, B, grad, T, tro, _capi.POOL_SUM, True, lr=0.0,
                                               workspace=ws[0])
        t_b = timeit(run, iters=5, warmup=1)
        t_f = timeit(lambda: ops.embbag_forward(w, idx, off, B, T, tro, _capi.POOL_SUM, True, out=o, out_ld=T * D),
                     iters=5, warmup=1)
        U = sum(int(torch.unique(idx[t * B:(t + 1) * B]).numel()) for t in range(T))
        rec = {"tables": name, "T": T, "unique": U, "bwd_us": t_b * 1e6, "fwd_us": t_f * 1e6,
               "bwd_GBps": (B * T * D * 4 + 2 * U * D * 4 + B * T * 4) / t_b / 1e9,
               "bwd_ns_per_position": t_b * 1e9 / (B * T)}
        out.append(rec)
        print(json.dumps(rec))
        del w
    return out


def interact(B=None, F=27, D=128):
    if B is None:
        return [interact(b, F, D) for b in (int(v) for v in os.environ.get("INTERACT_B", "65536,8192").split(","))]
    x = torch.randn(B, F, D, device="cuda")
    n_out = D + F * (F - 1) // 2
    ld = pad4(n_out)
    o = torch.empty(B, ld, device="cuda")
    do = torch.randn(B, ld, device="cuda")
    dx = torch.empty_like(x)
    t_f = timeit(lambda: ops.interact_dot_forward(x, out=o, out_ld=ld))
    t_b = timeit(lambda: ops.interact_dot_backward(x, do[:, :n_out], relu_mask_feature0=True, out=dx))
    rec = {"B": B, "fwd_us": t_f * 1e6, "fwd_GBps": B * (F * D + n_out) * 4 / t_f / 1e9, "bwd_us": t_b * 1e6,
           "bwd_GBps": B * (2 * F * D + n_out) * 4 / t_b / 1e9}
    print(json.dumps(rec))
    return rec


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    if os.environ.get("GEMM_MODE"):
        ops.set_gemm_mode(int(os.environ["GEMM_MODE"]))
    res = {"gemm": gemm, "gemm1024": lambda: gemm([(1024, 1024)]), "embbag": embbag, "sparse": sparse, "sparse_mlperf": sparse_mlperf,
           "interact": interact}[which]()
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)
