"""What do the wrong outputs of vip_window_attn_fwd_f16 under MFMA contention look like?  (follow-up of race_matrix.py)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import _abi, ops  # noqa: E402

g = torch.Generator().manual_seed(1)
ws = int(sys.argv[1]) if len(sys.argv) > 1 else 14
heads = 8 if ws == 14 else 2
B, nw = (64, 1) if ws == 14 else (16, 8)
C_ = heads * 32
qkv = (torch.randn(B, nw * ws, nw * ws, 3 * C_, generator=g)).to(torch.float16).cuda()
tab = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).cuda()
ref = ops.window_attention(qkv, None, tab, heads, ws, 32 ** -0.5).clone()
torch.cuda.synchronize()
sink = torch.zeros((16,), dtype=torch.float32, device="cuda")
lib = _abi.lib()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
flops = C.c_double(0.0)
shown = 0
for it in range(40):
    with torch.cuda.stream(sb):
        _abi.check(lib.vip_microbench_mfma_f16(sink.data_ptr(), 300, C.byref(flops), sb.cuda_stream), "mfma")
    with torch.cuda.stream(sa):
        out = ops.window_attention(qkv, None, tab, heads, ws, 32 ** -0.5)
    torch.cuda.synchronize()
    d = (out.float() - ref.float())
    bad = torch.nonzero(d.abs().amax(dim=-1) > 0)            # (b, y, x) pixels with any wrong channel
    if len(bad) == 0 or shown >= 3:
        continue
    shown += 1
    print(f"launch {it}: {len(bad)} wrong pixels; nan {int(torch.isnan(out.float()).sum())} inf {int(torch.isinf(out.float()).sum())}")
    b, y, x = bad[0].tolist()
    o, r = out[b, y, x].float().view(heads, 32), ref[b, y, x].float().view(heads, 32)
    for h in range(heads):
        if (o[h] != r[h]).any():
            ratio = (o[h] / r[h])
            print(f"  pixel (b {b}, y {y}, x {x}) token {(y % ws) * ws + x % ws} head {h}: wrong channels {int((o[h] != r[h]).sum())}/32; ratio out/ref "
                  f"min {float(ratio.min()):.4f} max {float(ratio.max()):.4f}; out[:6] {[round(v, 4) for v in o[h][:6].tolist()]} ref[:6] {[round(v, 4) for v in r[h][:6].tolist()]}")
    # which tokens of that (image, window, head) are wrong?
    wy, wx = y // ws, x // ws
    blk_o = out[b, wy * ws:(wy + 1) * ws, wx * ws:(wx + 1) * ws].float().reshape(ws * ws, heads, 32)
    blk_r = ref[b, wy * ws:(wy + 1) * ws, wx * ws:(wx + 1) * ws].float().reshape(ws * ws, heads, 32)
    for h in range(heads):
        toks = torch.nonzero((blk_o[:, h] != blk_r[:, h]).any(-1)).flatten().tolist()
        if toks:
            print(f"  item (b {b}, window {wy},{wx}, head {h}): wrong tokens {toks}")
            # is the wrong output a convex combination of V rows? compare with the softmax computed from wrong-scaled scores
            t0 = toks[0]
            print(f"     token {t0}: |out| {float(blk_o[t0, h].norm()):.3f} |ref| {float(blk_r[t0, h].norm()):.3f}  cos {float(torch.nn.functional.cosine_similarity(blk_o[t0, h], blk_r[t0, h], dim=0)):.4f}")
