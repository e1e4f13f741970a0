"""One fused-MLP shape, timed, for rocprofv3 passes: python tools/bench_mlp_one.py M C [n]   (hidden = 4 C, GELU, LN prologue, residual)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
M, C = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator().manual_seed(0)
x = torch.randn((M, C), generator=g).to("cuda", torch.float16)
fc1 = ops.make_dense_weight(torch.randn(C, 4 * C, generator=g) / C ** 0.5, torch.zeros(4 * C))
fc2 = ops.make_dense_weight(torch.randn(4 * C, C, generator=g) / (4 * C) ** 0.5, torch.zeros(C))
gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
ops.mlp(x, fc1, fc2, act="gelu", residual=x, ln=(gm, bt, 1e-6)); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.mlp(x, fc1, fc2, act="gelu", residual=x, ln=(gm, bt, 1e-6))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"mlp M={M} C={C}: {ms*1e3:.1f} us  {16.0*M*C*C/ms/1e9:.0f} TFLOP/s  {6.0*M*C/ms/1e6:.0f} GB/s")
