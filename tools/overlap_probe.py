"""Which neighbour disturbs the sparse row update?  The update (embbag_backward_apply, Criteo-Kaggle shape) runs on one
stream while an "aggressor" runs on a second one; the updated tables are compared bit for bit with a serial run.

  python tools/overlap_probe.py --aggressor wgrad|forward|forward_f32|dgrad|skinny|synthetic0..4|hipblas|stream|none [--reps 12]

Point DLRM_AMD_LIB at a build whose embbag_bwd.hip keeps packed float32 instructions to see the failure
(the shipped library is built without them and passes with every aggressor)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402

ROWS = [1460, 583, 1000000, 200000, 305, 24, 12517, 633, 3, 93145, 5683, 800000, 3194, 27, 14992, 500000, 10, 5652, 2173,
        4, 700000, 18, 15, 286181, 105, 142572]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aggressor", default="wgrad")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--launches", type=int, default=16)
    ap.add_argument("--victim", default="update", choices=["update", "synthetic", "synthetic-scalar", "synthetic-waw"],
                    help="synthetic: the bare load -> v_pk_mul_f32 / v_pk_fma_f32 pattern of tools/aggressors.hip, which checks itself")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, T, D = 2048, len(ROWS), 16
    tro = torch.tensor([0] + torch.tensor(ROWS).cumsum(0).tolist(), dtype=torch.int64, device=dev)
    w0 = torch.randn(int(tro[-1]), D, device=dev)
    idx = torch.cat([torch.randint(0, r, (B,), device=dev) for r in ROWS]).to(torch.int64)
    off = torch.arange(T * B, device=dev, dtype=torch.int64)
    grad = torch.randn(B, T * D, device=dev) * 1e-3
    dz = torch.randn(B, 512, device=dev)
    x = torch.randn(B, 367, device=dev)
    w = torch.randn(512, 367, device=dev)
    bias = torch.randn(512, device=dev)
    dw = torch.empty(512, 367, device=dev)
    db = torch.empty(512, device=dev)
    y = torch.empty(B, 512, device=dev)
    dx = torch.empty(B, 367, device=dev)
    big = torch.randn(64 << 20, device=dev)
    x13 = torch.randn(B, 13, device=dev)
    w13 = torch.randn(512, 13, device=dev)
    sink = torch.zeros(16, device=dev)
    agg = None
    if a.aggressor.startswith("synthetic"):
        agg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libaggressors.so"))
    if a.aggressor == "forward_f32":
        ops.set_gemm_mode(0)
    wsd = torch.empty(ops.linear_backward_weight_workspace_bytes(B, 512, 367), dtype=torch.uint8, device=dev)
    ws = ops.embbag_backward_prepare(w0, idx, off, B, num_tables=T, table_row_offsets=tro)

    def apply(t):
        ops.embbag_backward_apply(t, idx, off, B, grad, ws, num_tables=T, table_row_offsets=tro, lr=0.1)

    def aggress():
        if a.aggressor == "wgrad":
            ops.linear_backward_weight(dz, x, dw, db, workspace=wsd)
        elif a.aggressor == "forward":
            ops.linear_forward(x, w, bias, relu=True, out=y)
        elif a.aggressor == "dgrad":
            ops.linear_backward_data(dz, w, out=dx)
        elif a.aggressor == "hipblas":
            torch.matmul(dz.t(), x, out=dw)
        elif a.aggressor == "stream":
            big.mul_(1.0)
        elif a.aggressor == "forward_f32":      # the same GEMM kernel family on the float32 MFMA (no bf16 splits)
            ops.linear_forward(x, w, bias, relu=True, out=y)
        elif a.aggressor == "skinny":           # a layer with 13 inputs: no MFMA, no LDS tiles
            ops.linear_forward(x13, w13, bias, relu=True, out=y)
        elif a.aggressor.startswith("synthetic"):   # synthetic0..4: tools/aggressors.hip
            rc = agg.launch_aggressor(int(a.aggressor[9:]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                      ctypes.c_void_p(sink.data_ptr()), 1024, 4000)
            assert rc == 0


    if a.victim != "update":
        if agg is None:
            agg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libaggressors.so"))
        vec = torch.randn(1 << 22, device=dev) * 1e-3
        bad = torch.zeros(2, dtype=torch.int32, device=dev)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        bad_runs = 0
        for rep in range(a.reps):
            bad.zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(sb):
                for _ in range(a.launches if a.aggressor != "none" else 0):
                    aggress()
            with torch.cuda.stream(sa):
                rc = agg.launch_victim({"synthetic": 0, "synthetic-scalar": 1, "synthetic-waw": 2}[a.victim], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                       ctypes.c_void_p(vec.data_ptr()), 1 << 20, 100, ctypes.c_void_p(bad.data_ptr()))
                assert rc == 0
            torch.cuda.synchronize()
            lo, hi = bad.tolist()
            bad_runs += (lo + hi) > 0
            if lo + hi:
                print("   repetition %d: %d wrong low halves, %d wrong high halves" % (rep, lo, hi))
        print("victim %s, aggressor %-8s: %d of %d overlapped runs had wrong results" % (a.victim, a.aggressor, bad_runs, a.reps), flush=True)
        return
    ref = w0.clone()
    apply(ref)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad_runs = 0
    for rep in range(a.reps):
        t = w0.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(a.launches if a.aggressor != "none" else 0):
                aggress()
        with torch.cuda.stream(sa):
            apply(t)
        torch.cuda.synchronize()
        n = int((t != ref).any(dim=1).sum().item())
        bad_runs += n > 0
    print("aggressor %-8s: %d of %d overlapped runs differ from the serial update" % (a.aggressor, bad_runs, a.reps), flush=True)


if __name__ == "__main__":
    main()
