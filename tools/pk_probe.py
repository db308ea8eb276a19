"""The overlap scenario of tests/test_overlap_gpu.py on the diagnostic build (tools/build_pk_probe.sh: the sparse
update compiled with packed float32 instructions and an in-kernel probe that recomputes every accumulation with a
scalar fma).  One pass of 8 overlapped repetitions; prints, per repetition, the rows that differ from the serial
update and the probe's records:
  block, thread, position, element, gradient register, gradient re-read from memory, scale, accumulator before,
  packed result, scalar result

  python tools/pk_probe.py [out.json]          (run tools/build_pk_probe.sh first, on the CPU)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DLRM_AMD_LIB"] = os.path.join(ROOT, "dlrm_amd", "lib_probe", "libdlrm_amd.so")

import torch  # noqa: E402

from dlrm_amd import _capi, ops  # noqa: E402
from tests.test_overlap_gpu import ROWS  # noqa: E402


def read_probe():
    n = C.c_uint()
    rec = (C.c_uint * (12 * 256))()
    _capi.check(_capi.lib().dlrm_pk_probe_read(C.byref(n), rec, 256))
    return n.value, [list(rec[12 * i:12 * i + 12]) for i in range(min(n.value, 256))]


def main():
    gemm = int(os.environ.get("GEMM_MODE", "1"))      # bf16x3 GEMMs as neighbours, as when it was first seen
    ops.set_gemm_mode(gemm)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    B, T, D = 2048, len(ROWS), 16
    tro = torch.tensor([0] + torch.tensor(ROWS).cumsum(0).tolist(), dtype=torch.int64, device=dev)
    w0 = torch.randn(int(tro[-1]), D, device=dev, generator=gen)
    idx = torch.cat([torch.randint(0, r, (B,), device=dev, generator=gen) for r in ROWS]).to(torch.int64)
    off = torch.arange(T * B, device=dev, dtype=torch.int64)
    grad = torch.randn(B, T * D, device=dev, generator=gen) * 1e-3
    layers = []
    for N, K in ((256, 512), (512, 367)):
        layers.append((torch.randn(B, N, device=dev, generator=gen), torch.randn(B, K, device=dev, generator=gen),
                       torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
    wsd = torch.empty(max(ops.linear_backward_weight_workspace_bytes(B, dz.shape[1], x.shape[1]) for dz, x, _, _ in layers),
                      dtype=torch.uint8, device=dev)
    ws = ops.embbag_backward_prepare(w0, idx, off, B, num_tables=T, table_row_offsets=tro)

    def apply(w):
        ops.embbag_backward_apply(w, idx, off, B, grad, ws, num_tables=T, table_row_offsets=tro, lr=0.1)

    ref = w0.clone()
    apply(ref)
    torch.cuda.synchronize()
    out = {"gemm_mode": gemm, "serial_probe": read_probe()[0], "repetitions": []}
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(8):
        w = w0.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(8):
                for dz, x, dw, db in layers:
                    ops.linear_backward_weight(dz, x, dw, db, workspace=wsd)
        with torch.cuda.stream(sa):
            apply(w)
        torch.cuda.synchronize()
        bad = torch.nonzero((w != ref).any(dim=1)).flatten()
        n, recs = read_probe()
        diffs = []
        for r in bad[:8].tolist():
            cols = torch.nonzero(w[r] != ref[r]).flatten().tolist()
            diffs.append({"row": r, "elements": cols, "delta": [(w[r, c] - ref[r, c]).item() for c in cols]})
        out["repetitions"].append({"rows_differing": int(bad.numel()), "first": diffs, "probe_count": n,
                                   "probe_records": [["%d" % v if i < 4 or i >= 10 else "0x%08x" % v for i, v in enumerate(r)]
                                                     for r in recs[:16]]})
        print(json.dumps(out["repetitions"][-1]), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
