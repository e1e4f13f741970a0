"""Per-shape timing of the strict (packed) Dense / 1x1 convolution through ops.dense - whichever kernel the C dispatcher picks under the
current environment (VIP_G8P_MINK, VIP_PWK_XLK, VIP_PWK_WN2K, VIP_PW ...).  TF = logical flops.
    python tools/bench_gemm_h2.py [M N K act]..."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
SHAPES = [(147456, 1536, 384, "gelu"), (147456, 384, 1536, None), (50432, 1536, 384, "gelu"), (50432, 384, 1536, None), (50432, 1152, 384, None),
          (50432, 384, 384, None), (614656, 768, 192, "gelu"), (614656, 192, 768, None), (50176, 768, 256, "gelu"), (50176, 256, 768, None),
          (43264, 1024, 256, None), (43264, 1536, 384, None), (36864, 3072, 768, "gelu"), (36864, 768, 3072, None), (12544, 208, 1248, None)]
FAST = os.environ.get("VIP_BENCH_PREC", "strict") == "fast"      # the same shapes on the fp16 storage
if FAST:
    SHAPES = [(50176, 768, 256, "gelu"), (50176, 768, 256, None), (43264, 1024, 256, None), (50176, 256, 256, None), (50176, 512, 256, None),
              (160000, 512, 128, None), (43264, 512, 256, "silu"), (200704, 384, 128, None), (50176, 768, 192, None), (36864, 768, 192, None)]
_w = torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
tot = 0.0
for M, N, K, act in SHAPES:
    g = torch.Generator().manual_seed(0)
    pk = (lambda t: t.cuda().half()) if FAST else (lambda t: ops.pack_h2(t.cuda()))
    x = pk(torch.randn((M, K), generator=g))
    with ops.precision("fast" if FAST else "strict"):
        cw = ops.make_dense_weight(torch.randn(K, N, generator=g) / math.sqrt(K), torch.randn(N, generator=g) * 0.1)
    res = pk(torch.randn((M, N), generator=g)) if act is None else None
    y = ops.dense(x, cw, act=act, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dense(x, cw, act=act, residual=res)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    d = ops._abi.ConvDesc(B=1, H=1, W=M, Cin=K, Cout=N, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=1, Wo=M, groups=1, ldx=K, cin_off=0, ldy=N, cout_off=0,
                          ldr=N if res is not None else 0, res_off=0, ldw=cw.ldw, act_pre=ops._act(act), act_post=0)
    print(f"M={M:7d} N={N:5d} K={K:5d} {str(act):5s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF  {ops.conv_kernel_name(d, res is not None) if FAST else ops.conv_kernel_name_h2(d, res is not None)}", flush=True)
print(f"sum {tot:.3f} ms")
