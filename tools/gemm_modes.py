"""The three evaluations of the float32 layer products side by side: error against a float64 product and time per call.

  python tools/gemm_modes.py [M [out.json]]        (default M = 65536: the layers of the MLPerf-shape model)"""
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
    names = {0: "f32 MFMA", 1: "bf16x3", 2: "f16x2+pass", 3: "f16x2", 4: "f16x2+planes"}
    only = [int(v) for v in os.environ.get("GEMM_MODES_ONLY", "0,1,2,3,4").split(",")]
    records = []
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
        sw = ops.SplitWeights(w)
        for mode in only:      # 3: f16x2 with the operands' maxima handed in (no measuring passes); 4: and the weights pre-split
            ops.set_gemm_mode(min(mode, 2))
            kf = dict(x_absmax=ax, w_absmax=aw) if mode >= 3 else {}
            kd = dict(dz_absmax=adz, w_absmax=aw) if mode >= 3 else {}
            kw = dict(dz_absmax=adz, x_absmax=ax) if mode >= 3 else {}
            if mode == 4:
                kf["w_planes"] = sw
                kd["w_planes"] = sw
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
            records.append({"M": M, "N": N, "K": K, "evaluation": names[mode],
                            "forward_us": round(t_f, 1), "backward_data_us": round(t_d, 1), "backward_weight_us": round(t_w, 1),
                            "forward_tflops": round(fl / t_f / 1e6, 1), "backward_data_tflops": round(fl / t_d / 1e6, 1),
                            "backward_weight_tflops": round(fl / t_w / 1e6, 1),
                            "max_err_over_max_ref": {"forward": e_f, "backward_data": e_d, "backward_weight": e_w}})
            print("M=%d N=%4d K=%4d %-10s forward %7.1f us (%5.1f TF/s, err %.1e)  backward-data %7.1f us (%5.1f TF/s, err %.1e)  "
                  "backward-weight %7.1f us (%5.1f TF/s, err %.1e)" % (M, N, K, names[mode], t_f, fl / t_f / 1e6, e_f, t_d,
                                                                     fl / t_d / 1e6, e_d, t_w, fl / t_w / 1e6, e_w), flush=True)


    if len(sys.argv) > 2:
        import json
        json.dump({"command": "python tools/gemm_modes.py %d <this file>" % M,
                   "note": "one layer at a time, operands N(0,1) (x after ReLU, dz 1e-5 N(0,1) with half the entries "
                           "zero), 20 timed calls each; errors against float64 products (forward / backward-data on 2048 rows); "
                           "'f16x2+pass': the call measures both operands' maxima itself, 'f16x2': maxima handed in, "
                           "'f16x2+planes': and the weight matrix split into its f16 planes beforehand (dlrm_split_weights)",
                   "layers": records}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
