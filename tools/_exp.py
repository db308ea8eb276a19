import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops
def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for N, K in [(1024, 1024), (1024, 512), (512, 1024)]:
    x = torch.relu(torch.randn(M, K, device="cuda"))
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    dz = torch.randn(M, N, device="cuda") * (torch.rand(M, N, device="cuda") > 0.5)
    dx = torch.empty(M, K, device="cuda")
    for rep in range(2):
        a = timeit(lambda: ops.linear_backward_data(dz, w, relu_mask=x, out=dx))
        b = timeit(lambda: ops.linear_backward_data(dz, w, out=dx))
        print("N=%d K=%d dgrad with mask %.1f us, without %.1f us" % (N, K, a, b))
