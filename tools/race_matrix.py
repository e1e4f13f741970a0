"""Follow-up of tools/attn_race.py: WHICH co-resident kernels make vip_window_attn_fwd_f16 (and other kernels of this library) return
wrong results?  Victim on stream A, aggressor on stream B, every victim result compared bit for bit with its own solo result.

    python tools/race_matrix.py [--iters 40]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import _abi, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--victims", default="attn14,attn7,attn14g,mhsa,mlp_fused,mlp_stream,se_gate,dwconv3,dwconv7,layernorm,gemm,gemm8p,"
                                          "pw_stream,conv3x3,rows_gemm,gated,hilo,stem,sconv,sattn")
    ap.add_argument("--aggressors", default="mfma_only,pwk_plain,pw_stream,conv3x3,gemm8p")
    a = ap.parse_args()
    g = torch.Generator().manual_seed(1)

    def r(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).to(torch.float16).cuda()

    # ---- operands
    qkv14, tab14 = r(64, 14, 14, 768), (torch.randn(27 * 27, 8, generator=g) * 0.5).cuda()
    qkv7, tab7 = r(16, 56, 56, 192), (torch.randn(13 * 13, 2, generator=g) * 0.5).cuda()
    qkvm = r(32, 197, 3 * 384)
    xm = r(64 * 56 * 56, 96)
    f1 = ops.make_dense_weight(torch.randn(96, 384, generator=g) / 10, torch.zeros(384))
    f2 = ops.make_dense_weight(torch.randn(384, 96, generator=g) / 20, torch.zeros(96))
    xs = r(64, 28, 28, 192)
    s1 = ops.make_conv_weight(torch.randn(1, 1, 192, 8, generator=g) / 14, torch.zeros(8))
    s2 = ops.make_conv_weight(torch.randn(1, 1, 8, 192, generator=g) / 3, torch.zeros(192))
    xd3 = r(64, 56, 56, 64)
    wd3 = ops.make_dw_weight(torch.randn(3, 3, 64, 1, generator=g) / 3)
    xl = r(50176, 256)
    gam, bet = torch.ones(256).cuda(), torch.zeros(256).cuda()
    xg = r(50176, 256)
    cw256 = ops.make_dense_weight(torch.randn(256, 768, generator=g) / 16, torch.zeros(768))
    xg1k = r(50176, 1024)
    cw1k = ops.make_dense_weight(torch.randn(1024, 256, generator=g) / 32, torch.zeros(256))
    xps = r(64, 100, 100, 64)
    cwps = ops.make_conv_weight(torch.randn(1, 1, 64, 256, generator=g) / 8, torch.zeros(256))
    xc3 = r(64, 50, 50, 64)
    cwc3 = ops.make_conv_weight(torch.randn(3, 3, 64, 64, generator=g) / 24, torch.zeros(64))
    xd7 = r(64, 56, 56, 96)
    wd7 = ops.make_dw_weight(torch.randn(7, 7, 96, 1, generator=g) / 7)
    sink = torch.zeros((16,), dtype=torch.float32, device="cuda")
    src = torch.empty((1 << 27,), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    lib = _abi.lib()

    def st():
        return torch.cuda.current_stream().cuda_stream

    qg14 = r(64, 196, 256)
    qkv14g = r(64, 14, 14, 512)
    xm2 = r(64 * 28 * 28, 192)
    f1b = ops.make_dense_weight(torch.randn(192, 768, generator=g) / 14, torch.zeros(768))
    f2b = ops.make_dense_weight(torch.randn(768, 192, generator=g) / 28, torch.zeros(192))
    xr = r(256, 2048)
    cwr = ops.make_dense_weight(torch.randn(2048, 512, generator=g) / 45, torch.zeros(512))
    xgate = r(64, 14, 14, 960)
    gate = torch.stack([torch.rand(64, 960, generator=g).to(torch.float16), torch.zeros(64, 960, dtype=torch.float16)], 1).cuda().contiguous()
    cwgate = ops.make_conv_weight(torch.randn(1, 1, 960, 160, generator=g) / 31, torch.zeros(160))
    xh = r(64, 56, 56, 24)
    cwh = ops.make_conv_weight(torch.randn(1, 1, 24, 144, generator=g) / 5, torch.zeros(144), hilo=True)
    xstem = r(64, 200, 200, 8)
    cwstem = ops.make_conv_weight(torch.randn(3, 3, 8, 32, generator=g) / 8, torch.zeros(32))
    with ops.precision("strict"):
        cws = ops.make_conv_weight(torch.randn(1, 1, 256, 256, generator=g) / 16, torch.zeros(256))
    xs32 = torch.randn(64, 14, 14, 256, generator=g).cuda()
    qkvs = torch.randn(16, 14, 14, 768, generator=g).cuda()
    victims = {
        "attn14g": lambda: ops.window_attention(qkv14g, qg14, tab14, 8, 14, 32 ** -0.5),
        "mlp_stream": lambda: ops.mlp(xm2, f1b, f2b, act="gelu", residual=xm2),
        "dwconv7": lambda: ops.dwconv2d(xd7, wd7, None, 7, 1, (3, 3, 3, 3)),
        "gemm8p": lambda: ops.dense(xg1k, cw1k),
        "pw_stream": lambda: ops.conv2d(xps, cwps, act="relu"),
        "conv3x3": lambda: ops.conv2d(xc3, cwc3, pad=(1, 1, 1, 1), act="relu"),
        "rows_gemm": lambda: ops.dense(xr, cwr, act="relu"),
        "gated": lambda: ops.conv2d(xgate, cwgate, gate=gate),
        "hilo": lambda: ops.conv2d(xh, cwh, act="silu"),
        "stem": lambda: ops.conv2d(xstem, cwstem, stride=2, pad=(0, 1, 0, 1), act="silu"),
        "sconv": lambda: ops.conv2d(xs32, cws, act="gelu"),
        "sattn": lambda: ops.window_attention(qkvs, None, tab14, 8, 14, 32 ** -0.5),
        "attn14": lambda: ops.window_attention(qkv14, None, tab14, 8, 14, 32 ** -0.5),
        "attn7": lambda: ops.window_attention(qkv7, None, tab7, 2, 7, 32 ** -0.5),
        "mhsa": lambda: ops.mhsa(qkvm, 6, 0.125),
        "mlp_fused": lambda: ops.mlp(xm, f1, f2, act="gelu", residual=xm),
        "se_gate": lambda: ops.se_gate(xs, s1, s2, "silu", "sigmoid"),
        "dwconv3": lambda: ops.dwconv2d(xd3, wd3, None, 3, 1, (1, 1, 1, 1), act="gelu"),
        "layernorm": lambda: ops.layernorm(xl, gam, bet, 1e-5),
        "gemm": lambda: ops.dense(xg, cw256, act="gelu"),
    }
    flops = C.c_double(0.0)
    aggressors = {
        "pwk_gelu": lambda: [ops.dense(xg, cw256, act="gelu") for _ in range(4)],
        "pwk_plain": lambda: [ops.dense(xg, cw256) for _ in range(4)],
        "gemm8p": lambda: [ops.dense(xg1k, cw1k) for _ in range(4)],
        "pw_stream": lambda: [ops.conv2d(xps, cwps, act="relu") for _ in range(4)],
        "conv3x3": lambda: [ops.conv2d(xc3, cwc3, pad=(1, 1, 1, 1), act="relu") for _ in range(4)],
        "mfma_only": lambda: _abi.check(lib.vip_microbench_mfma_f16(sink.data_ptr(), 300, C.byref(flops), st()), "mfma"),
        "copy16": lambda: _abi.check(lib.vip_microbench_copy(src.data_ptr(), dst.data_ptr(), 1 << 27, st()), "copy"),
        "layernorm": lambda: [ops.layernorm(xl, gam, bet, 1e-5) for _ in range(6)],
        "dwconv7": lambda: [ops.dwconv2d(xd7, wd7, None, 7, 1, (3, 3, 3, 3)) for _ in range(2)],
        "attn14": lambda: [ops.window_attention(qkv14, None, tab14, 8, 14, 32 ** -0.5) for _ in range(4)],
    }
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    print(f"library: {_abi.LIB_PATH}")
    for vn in a.victims.split(","):
        vf = victims[vn]
        ref = vf()
        ref = (ref[0] if isinstance(ref, tuple) else ref).clone()
        torch.cuda.synchronize()
        row = []
        for an in a.aggressors.split(","):
            af = aggressors[an]
            bad, worst = 0, 0.0
            for _ in range(a.iters):
                with torch.cuda.stream(sb):
                    af()
                with torch.cuda.stream(sa):
                    out = vf()
                torch.cuda.synchronize()
                out = out[0] if isinstance(out, tuple) else out
                if not torch.equal(out, ref):
                    bad += 1
                    worst = max(worst, float((out.float() - ref.float()).abs().max()))
            row.append(f"{an}:{bad}/{a.iters}" + (f"(max {worst:.1e})" if bad else ""))
        print(f"[race2] victim {vn:10s} | " + "  ".join(row))


if __name__ == "__main__":
    main()
