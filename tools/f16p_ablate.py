"""Times one ablated variant of gemm_f16p_kernel (DLRM_F16P_ABLATE=<mask> in the environment, see gemm_f32.hip) on the
forward products of the MLPerf-shape layers at batch 65536.  One process per variant: the mask is read once.

  DLRM_F16P_ABLATE=<mask> python tools/f16p_ablate.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DLRM_AMD_LIB"] = os.path.join(ROOT, "dlrm_amd", "lib_ablate", "libdlrm_amd.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dlrm_amd import ops  # noqa: E402
from tools.gemm_modes import timeit  # noqa: E402

M = 65536
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ops.set_gemm_mode(2)
out = {"mask": int(os.environ.get("DLRM_F16P_ABLATE", "0")), "a_pad": os.environ.get("A_PAD"), "y_pad": os.environ.get("Y_PAD"),
       "plane_pad": os.environ.get("DLRM_PLANE_PAD"), "us": {}}
for N, K in ((1024, 1024), (512, 1024), (1024, 512)):
    x = torch.relu(torch.randn(M, K + int(os.environ.get("A_PAD", "0")), device=dev))[:, :K]
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(M, N + int(os.environ.get("Y_PAD", "0")), device=dev)[:, :N]
    ax = x.abs().max().reshape(1)
    sw = ops.SplitWeights(w)
    us = timeit(lambda: ops.linear_forward(x, w, None, relu=False, out=y, x_absmax=ax, w_planes=sw))
    out["us"]["%dx%d" % (N, K)] = round(us, 1)
    out.setdefault("mfma_frac", {})["%dx%d" % (N, K)] = round(3 * 2.0 * M * N * K / (us * 1e-6) / 2.5e15, 3)
print(json.dumps(out))
