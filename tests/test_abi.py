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


def test_missing_library_fails_loudly(monkeypatch):
    from vipcup_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libvipcup_hip.so")
    with pytest.raises(_abi.VipError):
        _abi.lib()
