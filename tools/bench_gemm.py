"""Micro-benchmark of the dense/1x1 GEMM path on the dominant ensemble shapes (M,N,K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
SHAPES = [(2509056, 384, 96), (2509056, 96, 384), (614656, 768, 192), (614656, 192, 768), (147456, 1536, 384),
          (147456, 384, 1536), (36864, 3072, 768), (36864, 768, 3072), (50176, 768, 256), (50176, 256, 768),
          (802816, 192, 64), (802816, 64, 192), (640000, 256, 64), (640000, 64, 256), (160000, 512, 128),
          (200704, 384, 128), (43264, 1024, 256), (12544, 1536, 512), (3211264, 144, 24)]
if "--deep" in sys.argv:      # the compute-bound shapes only (K >= 384)
    SHAPES = [s for s in SHAPES if s[2] >= 384] + [(43264, 1536, 384), (43264, 256, 1024), (12544, 2048, 512), (50176, 768, 3072)]
for M, N, K in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((M, K), generator=g, dtype=torch.float16).cuda() if M * K < 3e8 else torch.randn((M, K), dtype=torch.float16, device="cuda")
    cw = ops.make_dense_weight(torch.randn((K, N), generator=g) * 0.05, torch.zeros(N))
    res = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    ACT = None if "--noact" in sys.argv else "gelu"
    y = ops.dense(x, cw, act=ACT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dense(x, cw, act=ACT)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M:8d} N={N:5d} K={K:5d} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF {2.0*(M*K+M*N)/ms/1e6:7.0f} GB/s", flush=True)
    del x, y, res
