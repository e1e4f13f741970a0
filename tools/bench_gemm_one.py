"""One dense shape, a few launches (for rocprofv3 --pmc runs):  python tools/bench_gemm_one.py M N K [act] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
act = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "none" else None
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
STRICT = os.environ.get("VIP_BENCH_PREC", "fast") == "strict"        # the packed storage (h2 kernels)
if STRICT:
    x = ops.pack_h2(torch.randn((M, K), device="cuda"))
    with ops.precision("strict"):
        cw = ops.make_dense_weight(torch.randn((K, N)) * 0.05, torch.zeros(N))
else:
    x = torch.randn((M, K), dtype=torch.float16, device="cuda")
    cw = ops.make_dense_weight(torch.randn((K, N)) * 0.05, torch.zeros(N))
for _ in range(2):
    ops.dense(x, cw, act=act)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.dense(x, cw, act=act)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"M={M} N={N} K={K} act={act} {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF {2.0*(M*K+M*N)/ms/1e6:.0f} GB/s")
