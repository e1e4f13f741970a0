"""CPU: the C-ABI library builds, loads, and exports every symbol include/vipcup_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vipcup_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("vip_conv2d_nhwc_f16", "vip_window_attn_fwd_f16", "vip_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import _abi, build
    build.build_lib()
    lib = _abi.lib()
    for name in _declared():
        assert hasattr(lib, name), f"libvipcup_hip.so does not export {name}"
        assert name in _abi.SIGNATURES, f"_abi.SIGNATURES lacks {name}"
    assert lib.vip_version() >= 1000


def test_argument_checks_do_not_need_a_gpu():
    """Bad arguments are rejected before any HIP call, with a message."""
    import ctypes as C
    from vipcup_amd import _abi
    lib = _abi.lib()
    d = _abi.ConvDesc()
    st = lib.vip_conv2d_nhwc_f16(None, None, None, None, None, C.byref(d), None)
    assert st == -1 and b"null" in lib.vip_last_error()
    st = lib.vip_layernorm_f16(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 4, 12, 1e-5, None)
    assert st == -2
    p = C.c_void_p(16)
    # fused MLP: shape support is a pure function; unsupported shapes / half-given LayerNorm are refused with a message
    assert lib.vip_mlp_fused_supported(100000, 96, 384, 3) == 1 and lib.vip_mlp_fused_supported(100000, 192, 768, 3) == 1
    assert lib.vip_mlp_fused_supported(100000, 96, 384, 2) == 0      # only GELU
    assert lib.vip_mlp_fused_supported(100000, 384, 1536, 3) == 0    # weights do not fit / y accumulators too wide
    assert lib.vip_mlp_fused_supported(100, 96, 384, 3) == 0         # too few tokens to fill the chip
    st = lib.vip_mlp_fused_f16(p, None, None, 0.0, p, None, p, None, None, p, 100000, 384, 1536, 384, 384, 1536, 384, 0, 3, None)
    assert st == -3 and b"unsupported" in lib.vip_last_error()
    st = lib.vip_mlp_fused_f16(p, p, None, 1e-6, p, None, p, None, None, p, 100000, 96, 384, 96, 96, 384, 96, 0, 3, None)
    assert st == -1 and b"ln_gamma" in lib.vip_last_error()
    # SE gate: alignment and width limits
    st = lib.vip_se_gate_f16(p, p, None, p, None, p, 4, 49, 100, 100, 8, 104, 100, 8, 2, 4, 1, None)
    assert st == -2
    st = lib.vip_se_gate_f16(p, p, None, p, None, p, 4, 49, 32768, 32768, 8, 32768, 32768, 8, 2, 4, 1, None)
    assert st == -3 and b"too wide" in lib.vip_last_error()
    # split-output Dense: a batch of pooled vectors, at most 256 rows
    st = lib.vip_gemm_split2_f16(p, p, None, p, 257, 64, 64, 64, 4, None)
    assert st == -3 and b"256" in lib.vip_last_error()
    assert lib.vip_global_avgpool_split_f16(p, p, 2, 4, 12, 12, None) == -2
    assert lib.vip_gap_ln_dense_f32(p, p, p, 1e-6, p, None, p, 2, 4, 8192, 8192, 1, None) == -3
    st = lib.vip_gemm_split_f16(p, p, None, p, 257, 64, 64, 64, 64, 4, None)
    assert st == -3 and b"256" in lib.vip_last_error()
    st = lib.vip_mul_f16(p, p, p, 4, 16, 32, 24, 32, 0, 16, 0, None)          # slice 24..40 of a 32-wide row
    assert st == -1 and b"exceeds" in lib.vip_last_error()
    st = lib.vip_mul_f16(p, p, p, 4, 12, 32, 0, 32, 0, 16, 0, None)
    assert st == -2
    st = lib.vip_scale_add_act3_f16(p, p, 3, None, p, None, 1, 4, 8, 0, 0, None)
    assert st == -1 and b"scale_planes" in lib.vip_last_error()
    # two-term weights: pointwise, K <= 256 only
    d = _abi.ConvDesc(B=1, H=8, W=8, Cin=512, Cout=16, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=8, Wo=8, groups=1, ldx=512,
                      cin_off=0, ldy=16, cout_off=0, ldr=0, res_off=0, ldw=512, act_pre=0, act_post=0)
    st = lib.vip_conv2d_hilo_nhwc_f16(p, p, p, None, None, p, C.byref(d), None)
    assert st == -3 and b"K <= 256" in lib.vip_last_error()
    # gated conv: only pointwise convolutions take a gate
    d = _abi.ConvDesc(B=1, H=8, W=8, Cin=16, Cout=16, kh=3, kw=3, sh=1, sw=1, pt=1, pl=1, Ho=8, Wo=8, groups=1, ldx=16,
                      cin_off=0, ldy=16, cout_off=0, ldr=0, res_off=0, ldw=144, act_pre=0, act_post=0)
    st = lib.vip_conv2d_gated_nhwc_f16(p, p, p, None, None, p, C.byref(d), None)
    assert st == -3 and b"gate" in lib.vip_last_error()
    st = lib.vip_conv2d_gated_nhwc_f16(p, None, p, None, None, p, C.byref(d), None)
    assert st == -1


def test_packed_strict_entry_points_check_arguments_without_a_gpu():
    """the round-4 entry points of the packed strict storage: shape support is a pure function, bad arguments are refused with a message"""
    import ctypes as C
    from vipcup_amd import _abi
    lib = _abi.lib()
    p = C.c_void_p(64)
    # fused MLP: C = 64 / 96 / 128, GELU, hidden % 32 == 0, enough tokens
    assert lib.vip_mlp_fused_supported_h2(100000, 96, 384, 3) == 1 and lib.vip_mlp_fused_supported_h2(100000, 128, 512, 3) == 1
    assert lib.vip_mlp_fused_supported_h2(100000, 192, 768, 3) == 0 and lib.vip_mlp_fused_supported_h2(100000, 96, 384, 2) == 0
    assert lib.vip_mlp_fused_supported_h2(100, 96, 384, 3) == 0 and lib.vip_mlp_fused_supported_h2(100000, 96, 400, 3) == 0
    st = lib.vip_mlp_fused_h2(p, None, None, 0.0, p, None, 1.0, p, None, 1.0, None, p, 100000, 192, 768, 192, 384, 1536, 192, 0, 3, None, None)
    assert st == -3 and b"unsupported" in lib.vip_last_error()
    st = lib.vip_mlp_fused_h2(p, p, None, 1e-6, p, None, 1.0, p, None, 1.0, None, p, 100000, 96, 384, 96, 192, 768, 96, 0, 3, None, None)
    assert st == -1 and b"ln_gamma" in lib.vip_last_error()
    st = lib.vip_mlp_fused_h2(p, None, None, 0.0, p, None, 0.0, p, None, 1.0, None, p, 100000, 96, 384, 96, 192, 768, 96, 0, 3, None, None)
    assert st == -1 and b"out_scale" in lib.vip_last_error()
    st = lib.vip_mlp_fused_h2(p, None, None, 0.0, p, None, 1.0, p, None, 1.0, None, p, 100000, 96, 384, 96, 100, 768, 96, 0, 3, None, None)
    assert st == -2                                                  # weight rows of 16 halfs
    # LDS-staged depthwise: k = 3 / 5 / 7, C % 8 == 0, tensors below 4 GiB
    assert lib.vip_dwconv2d_s1_supported_h2(256, 99, 99, 96, 7, 99, 99) == 1 and lib.vip_dwconv2d_s1_supported_h2(256, 7, 7, 1632, 5, 7, 7) == 1
    assert lib.vip_dwconv2d_s1_supported_h2(256, 99, 99, 96, 4, 99, 99) == 0 and lib.vip_dwconv2d_s1_supported_h2(256, 99, 99, 100, 7, 99, 99) == 0
    assert lib.vip_dwconv2d_s1_supported_h2(4096, 200, 200, 96, 7, 200, 200) == 0          # 4 B x 15.7 G elements
    st = lib.vip_dwconv2d_s1_h2(p, p, None, p, 2, 14, 14, 32, 9, 4, 4, 14, 14, 0, None, None)
    assert st == -3 and b"k = 3 / 5 / 7" in lib.vip_last_error()
    st = lib.vip_dwconv2d_s1_h2(p, None, None, p, 2, 14, 14, 32, 3, 1, 1, 14, 14, 0, None, None)
    assert st == -1
    st = lib.vip_dw_filter_quad_major(p, p, 3, 30, None)
    assert st == -1 and b"multiple of 4" in lib.vip_last_error()
    # gated conv on the packed storage: pointwise only, the gate spans the whole input channel axis
    d = _abi.ConvDesc(B=1, H=8, W=8, Cin=16, Cout=16, kh=3, kw=3, sh=1, sw=1, pt=1, pl=1, Ho=8, Wo=8, groups=1, ldx=16,
                      cin_off=0, ldy=16, cout_off=0, ldr=0, res_off=0, ldw=288, act_pre=0, act_post=0)
    st = lib.vip_conv2d_gated_nhwc_h2(p, p, p, None, None, p, C.byref(d), 1.0, None, None)
    assert st == -3 and b"gate" in lib.vip_last_error()
    st = lib.vip_conv2d_gated_nhwc_h2(p, None, p, None, None, p, C.byref(d), 1.0, None, None)
    assert st == -1
    d = _abi.ConvDesc(B=1, H=8, W=8, Cin=16, Cout=16, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=8, Wo=8, groups=1, ldx=32,
                      cin_off=0, ldy=16, cout_off=0, ldr=0, res_off=0, ldw=32, act_pre=0, act_post=0)
    st = lib.vip_conv2d_gated_nhwc_h2(p, p, p, None, None, p, C.byref(d), 1.0, None, None)
    assert st == -3 and b"ldx = Cin" in lib.vip_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from vipcup_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libvipcup_hip.so")
    with pytest.raises(_abi.VipError):
        _abi.lib()


def test_product_fails_loudly_without_library_or_gpu(monkeypatch):
    """No CPU fallback anywhere: a missing .so raises on first use, and the CLI / bench refuse to start without a GPU."""
    import subprocess
    import sys
    import pytest
    import torch
    import vipcup_amd  # noqa: F401
    from vipcup_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", os.path.join(os.path.dirname(_abi.LIB_PATH), "no_such_library.so"))
    with pytest.raises(_abi.VipError, match="no CPU fallback"):
        _abi.lib()
    monkeypatch.undo()
    assert _abi.lib() is not None
    if not torch.cuda.is_available():
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for cmd in ([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                    [sys.executable, os.path.join(root, "vip-cup-2022_amd", "main.py"), "in.csv", "out.csv", "--synthetic"]):
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=300)
            assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout), (cmd, r.stderr[-500:])
