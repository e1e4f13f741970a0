"""How accurate is a packed-storage product whose two SMALL terms run in block-scaled fp8?   (CPU emulation, numpy)

Strict mode computes  w x ~= w_hi x_hi + (w_lo x_hi + w_hi x_lo)  with three fp16 MFMAs per fragment pair.  The bracket is 2^-11 of the
product, so it needs ~11 bits relative to ITSELF to keep 2^-22 overall - DESIGN.md section 8.1 proposes to run it on
v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 values, one power-of-two scale per 32 consecutive k, fp32 accumulate) at twice the fp16 rate.
This script emulates that arithmetic on random layers and reports the error of each scheme against float64:
    python tools/micro/fp8_small_terms_emulation.py
"""
import numpy as np


def rn16(v):
    return v.astype(np.float16).astype(np.float64)


def e4m3(v):
    """round to OCP fp8 e4m3 (bias 7, max 448, subnormals, no inf) - values, not bit patterns"""
    a = np.abs(v)
    out = np.zeros_like(v)
    nz = a > 0
    e = np.floor(np.log2(a[nz]))
    e = np.clip(e, -6, 8)                       # normal exponents -6 .. 8; below: subnormal step 2^-9
    step = 2.0 ** (e - 3)
    q = np.round(a[nz] / step) * step
    q = np.minimum(q, 448.0)
    out[nz] = np.sign(v[nz]) * q
    return out


def block_fp8(v, axis_len=32):
    """per 32 consecutive k: scale = 2^ceil(log2(max|v| / 448)) (E8M0), values e4m3(v / scale)"""
    shp = v.shape
    b = v.reshape(shp[0], -1, axis_len)
    m = np.abs(b).max(axis=2, keepdims=True)
    s = np.where(m > 0, 2.0 ** np.ceil(np.log2(np.maximum(m, 1e-300) / 448.0)), 1.0)
    return (e4m3(b / s) * s).reshape(shp)


def run(M, K, N, seed):
    g = np.random.default_rng(seed)
    x = g.standard_normal((M, K)) * np.exp(g.standard_normal((M, 1)) * 0.5)       # rows of different scale
    w = g.standard_normal((N, K)) / np.sqrt(K)
    xh = rn16(x); xl = rn16(x - xh)
    wh = rn16(w); wl = rn16(w - wh)
    exact = x @ w.T
    ref = np.abs(exact).max()
    three = xh @ wh.T + xh @ wl.T + xl @ wh.T                                  # fp64 accumulation: isolates the representation error
    two_hi_only = xh @ wh.T
    fp8 = xh @ wh.T + block_fp8(xh) @ block_fp8(wl).T + block_fp8(xl) @ block_fp8(wh).T
    fp16_in = rn16(x) @ rn16(w).T
    return {k: np.abs(v - exact).max() / ref for k, v in
            (("three fp16 MFMAs (strict today)", three), ("main term + two fp8 small terms", fp8), ("main term only (= fp16 operands)", two_hi_only))}


if __name__ == "__main__":
    for (M, K, N) in ((512, 384, 256), (512, 1536, 384), (512, 96, 384), (256, 2304, 256)):
        r = run(M, K, N, seed=K)
        print(f"M={M} K={K} N={N}: max |err| / max |y| - " + "; ".join(f"{k}: {v:.2e}" for k, v in r.items()))
