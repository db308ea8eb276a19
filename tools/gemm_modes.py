"""The three evaluations of the float32 layer products side by side: error against a float64 product and time per call.

  python tools/gemm_modes.py [M]        (default M = 65536: the layers of the MLPerf-shape model)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    names = {0: "f32 MFMA", 1: "bf16x3", 2: "f16x2+pass", 3: "f16x2"}
    for N, K in ((1024, 479), (1024, 1024), (512, 1024), (256, 512)):
        Kp = (K + 3) // 4 * 4
        x = torch.relu(torch.randn(M, Kp, device=dev))[:, :K]
        w = (torch.randn(N, Kp, device=dev) / K ** 0.5)[:, :K]
        b = torch.randn(N, device=dev) * 0.1
        dz = torch.randn(M, N, device=dev) * 1e-5 * (torch.rand(M, N, device=dev) > 0.5)
        y = torch.empty(M, N, device=dev)
        dx = torch.empty(M, Kp, device=dev)[:, :K]
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        rows = slice(0, 2048)        # float64 references on a slice of the batch (the weight gradient on all of it)
        y64 = torch.relu(x[rows].double() @ w.double().t() + b.double())
        dx64 = (dz[rows].double() @ w.double()) * (x[rows] > 0)
        dw64 = dz.double().t() @ x.double()
        ax, aw, adz = x.abs().max().reshape(1), w.abs().max().reshape(1), dz.abs().max().reshape(1)
        for mode in (0, 1, 2, 3):      # 3: f16x2 with the operands' maxima handed in (no measuring passes)
            ops.set_gemm_mode(min(mode, 2))
            kf = dict(x_absmax=ax, w_absmax=aw) if mode == 3 else {}
            kd = dict(dz_absmax=adz, w_absmax=aw) if mode == 3 else {}
            kw = dict(dz_absmax=adz, x_absmax=ax) if mode == 3 else {}
            t_f = timeit(lambda: ops.linear_forward(x, w, b, relu=True, out=y, **kf))
            e_f = ((y[rows].double() - y64).abs().max() / y64.abs().max()).item()
            t_d = timeit(lambda: ops.linear_backward_data(dz, w, relu_mask=x, out=dx, **kd))
            e_d = ((dx[rows].double() - dx64).abs().max() / dx64.abs().max()).item()
            ws = [None]

            def wg():
                _, _, ws[0] = ops.linear_backward_weight(dz, x, dw, db, workspace=ws[0], **kw)
            t_w = timeit(wg)
            e_w = ((dw.double() - dw64).abs().max() / dw64.abs().max()).item()
            fl = 2.0 * M * N * K
            print("M=%d N=%4d K=%4d %-10s forward %7.1f us (%5.1f TF/s, err %.1e)  backward-data %7.1f us (%5.1f TF/s, err %.1e)  "
                  "backward-weight %7.1f us (%5.1f TF/s, err %.1e)" % (M, N, K, names[mode], t_f, fl / t_f / 1e6, e_f, t_d,
                                                                     fl / t_d / 1e6, e_d, t_w, fl / t_w / 1e6, e_w), flush=True)


if __name__ == "__main__":
    main()
