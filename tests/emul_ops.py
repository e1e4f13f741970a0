"""Test infrastructure: CPU emulation of vipcup_amd.ops on top of the oracle primitives, so the product's host
graphs (with their FOLDED, fp16-rounded weights) can be executed in fp32 on the CPU.  Two knobs separate the
error sources of the HIP path:  round_act  — round every op output to fp16 (what the kernels store);
                                weights are always the product's fp16 tensors.
Usage:  with emul_ops.patched(round_act=False): model = spec.ctor(params_on_cpu) ; model.logits(x8)"""
import contextlib

import torch

from oracle import ops_ref as R

ROUND_ACT = False
EXACT_W = False          # diagnostic: use the fp32 weights (stored fp16 + kept rounding error)
BIAS_CORRECT = False     # add (W32 - W16) . E[x] to every conv/dense output (what calibrated bias correction does)
ACTN = {0: None, 1: "relu", 2: "silu", 3: "gelu", 4: "sigmoid", None: None}


SKIP_ROUND = set()       # diagnostic: operator tags whose outputs are NOT rounded (conv, conv_res, dense, se_hid, se_gate, dw, saa, gap, gate_mul, ln, pool, attn)


SKIP_SMALL = 0           # diagnostic: feature maps with H <= SKIP_SMALL (and pooled vectors) are NOT rounded


def _r(t, tag="other"):
    if SKIP_SMALL and (t.dim() < 4 or t.shape[1] <= SKIP_SMALL):      # SKIP_SMALL = -1: only the pooled vectors / gates (no feature map)
        return t
    return t.to(torch.float16).to(torch.float32) if (ROUND_ACT and tag is not None and tag not in SKIP_ROUND) else t


def _an(a):
    return ACTN.get(a, a) if not isinstance(a, str) else a


def _w(cw):
    from vipcup_amd import ops
    k = cw.kh * cw.kw * cw.cin_g
    if ops._EXACT and cw.exact is not None:               # ops.exact_weights(): the K-doubled twin [w | lo] per tap and group
        w2 = cw.exact.w.float()[:, :2 * k].reshape(cw.cout, cw.kh * cw.kw, 2, cw.cin_g)
        return (w2[:, :, 0] + w2[:, :, 1]).reshape(cw.cout, cw.kh, cw.kw, cw.cin_g).permute(1, 2, 3, 0)
    w = cw.w.float() if getattr(cw, "w_lo", None) is None else cw.w.float() + cw.w_lo.float()     # two-term weights
    if EXACT_W and cw.err is not None:
        w = w + cw.err
    return w[:, :k].reshape(cw.cout, cw.kh, cw.kw, cw.cin_g).permute(1, 2, 3, 0)


def _b(cw):
    from vipcup_amd import ops
    return cw.exact.bias if (ops._EXACT and cw.exact is not None) else cw.bias


def conv2d(x, cw, stride=1, pad=(0, 0, 0, 0), act=None, act_post=None, residual=None, out=None, cin_off=0, cout_off=0,
           gate=None):
    xx = x[..., cin_off:cin_off + cw.cin]
    if gate is not None:
        xx = _r(xx * _gate(gate)[:, None, None, :], "gate_mul")
    _calib(cw, xx)
    y = R.conv2d(xx, _w(cw), _b(cw), stride, pad, cw.groups)
    if BIAS_CORRECT and not EXACT_W and cw.err is not None:
        k = cw.kh * cw.kw * cw.cin_g
        e = cw.err[:, :k].reshape(cw.groups, cw.cout // cw.groups, cw.kh * cw.kw, cw.cin_g)
        if TAP_MEANS:
            mu = tap_means(xx, cw.kh, cw.kw, stride, pad, y.shape[1], y.shape[2]).reshape(cw.kh * cw.kw, cw.groups, cw.cin_g)
            corr = torch.einsum("gotc,tgc->go", e, mu).reshape(cw.cout)
        else:
            mu = xx.reshape(-1, xx.shape[-1]).mean(0).reshape(cw.groups, cw.cin_g)
            corr = torch.einsum("gotc,gc->go", e, mu).reshape(cw.cout)
        y = y + corr
    y = R.act(y, _an(act))
    if residual is not None:
        y = y + residual[..., :cw.cout]
    y = _r(R.act(y, _an(act_post)), "conv_res" if residual is not None else "conv")
    if out is not None:
        out[..., cout_off:cout_off + cw.cout] = y
        return out
    return y


TAP_MEANS = False


def tap_means(xx, kh, kw, stride, pad, Ho, Wo):
    """[kh*kw, Cin]: mean over images and OUTPUT positions of the input value each filter tap sees (zero where it falls in the padding)"""
    import torch.nn.functional as F
    st = (stride, stride) if isinstance(stride, int) else stride
    xp = F.pad(xx, (0, 0, pad[2], pad[3], pad[0], pad[1]))
    out = []
    for i in range(kh):
        for j in range(kw):
            out.append(xp[:, i:i + st[0] * (Ho - 1) + 1:st[0], j:j + st[1] * (Wo - 1) + 1:st[1], :].reshape(-1, xx.shape[-1]).mean(0))
    return torch.stack(out)


def _calib(cw, xx):
    """inside the product's ops.calibration(): fold (W32 - W16) . E[x] into the bias, exactly as ops.conv2d / ops.dense do"""
    from vipcup_amd import ops
    if ops._CALIB and cw.err is not None:
        ops._bias_correct(cw, xx)


def dense(x, cw, act=None, act_post=None, residual=None, tag="dense"):
    _calib(cw, x)
    y = x @ _w(cw)[0, 0] + (_b(cw) if _b(cw) is not None else 0)
    if BIAS_CORRECT and not EXACT_W and cw.err is not None:
        y = y + cw.err[:, :x.shape[-1]] @ x.reshape(-1, x.shape[-1]).mean(0)
    y = R.act(y, _an(act))
    if residual is not None:
        y = y + residual
    return _r(R.act(y, _an(act_post)), "stream" if (residual is not None and tag == "dense") else tag)


def mlp(x, fc1, fc2, act="gelu", residual=None, ln=None):
    if ln is not None:
        x = layernorm(x, ln[0], ln[1], float(ln[2]))
    return dense(dense(x, fc1, act=act), fc2, residual=residual)


def _gate(g):
    """[B, 2, C] split gate (or plain [B, C]) -> fp32 [B, C]"""
    return g[:, 0] + g[:, 1] if g.dim() == 3 else g


def _split(v):
    if not ROUND_ACT or "se_gate" in SKIP_ROUND:
        return torch.stack([v, torch.zeros_like(v)], 1)
    hi = v.to(torch.float16).to(torch.float32)
    return torch.stack([hi, (v - hi).to(torch.float16).to(torch.float32)], 1)


def dense_split(x, cw, act=None):
    """[M, K] or split [M, 2, K] rows -> split [M, 2, N]"""
    return _split(dense(_gate(x), cw, act=act, tag=None))


def se_gate(x, fc1, fc2, act1, act2="sigmoid", split=True):
    # the fused kernel keeps the pooled and hidden vectors in fp32; the wide-gate path carries them as hi/lo planes
    fused = x.shape[-1] * fc1.cout + fc1.cout * fc2.cout <= 256 * 1024
    if fused:
        pooled = x.reshape(x.shape[0], -1, x.shape[-1]).mean(1)
        g = _split(dense(dense(pooled, fc1, act=act1, tag=None), fc2, act=act2, tag=None))
    else:
        g = dense_split(dense_split(global_avgpool(x, split=True), fc1, act=act1), fc2, act=act2)
    return g if split else g[:, 0]


def dwconv2d(x, w_khwc, bias, k, stride=1, pad=(0, 0, 0, 0), act=None):
    return _r(R.act(R.dwconv2d(x, w_khwc.float()[..., None], bias, stride, pad), _an(act)), "dw")


def dwconv2d_se(x, w_khwc, bias, k, stride, pad, act, fc1, fc2, act1, act2="sigmoid", split=True):
    # the pooling form sums the UNROUNDED fp32 outputs of the depthwise kernel (stride 1, small gate); the fallback pools the stored map
    raw = R.act(R.dwconv2d(x, w_khwc.float()[..., None], bias, stride, pad), _an(act))
    h = _r(raw, "dw")
    small = x.shape[-1] * fc1.cout + fc1.cout * fc2.cout <= 256 * 1024
    if stride == 1 and small and k in (3, 5, 7):
        pooled = raw.reshape(raw.shape[0], -1, raw.shape[-1]).mean(1)
        g = _split(dense(dense(pooled, fc1, act=act1, tag=None), fc2, act=act2, tag=None))
        return h, (g if split else g[:, 0])
    return h, se_gate(h, fc1, fc2, act1, act2, split)


def mbconv_expand_dw(x, cw, w_khwc, dw_bias, k, stride, pad, act=None):
    return dwconv2d(conv2d(x, cw, act=act), w_khwc, dw_bias, k, stride, pad, act=act)


def layernorm(x, gamma, beta, eps):
    return _r(R.layernorm(x, gamma, beta, eps), "ln")


POOL_MAX_ZEROPAD, POOL_AVG_VALID, POOL_AVG_FULL = 0, 1, 2


def pool2d(x, k, stride, pad=(0, 0, 0, 0), mode=0, out_hw=None):
    if out_hw is not None:
        return pool2d(x, k, stride, pad, mode)[:, :out_hw[0], :out_hw[1], :].contiguous()
    if mode == 0:
        return _r(R.maxpool_valid(x, k, stride, pad), "pool")
    if mode == 2:
        return _r(R.avgpool_valid(x, k, stride, pad), "pool")
    xp = R.zero_pad(x, pad).permute(0, 3, 1, 2)
    ones = R.zero_pad(torch.ones_like(x[..., :1]), pad).permute(0, 3, 1, 2)
    F = torch.nn.functional
    return _r((F.avg_pool2d(xp, k, stride) / F.avg_pool2d(ones, k, stride)).permute(0, 2, 3, 1).contiguous(), "pool")


def global_avgpool(x, split=False):
    B, C = x.shape[0], x.shape[-1]
    v = x.reshape(B, -1, C).mean(1)
    return _split(v) if split else _r(v, "gap")


def gap_ln_dense_f32(x, gamma, beta, eps, w_nc, bias):
    B, C = x.shape[0], x.shape[-1]
    v = R.layernorm(x.reshape(B, -1, C).mean(1), gamma, beta, eps)
    return v @ w_nc.t() + (bias if bias is not None else 0)


def gap_dense_f32(x, w_nc, bias):
    B, C = x.shape[0], x.shape[-1]
    return x.reshape(B, -1, C).mean(1) @ w_nc.t() + (bias if bias is not None else 0)


def head_prob(z):
    return torch.sigmoid(z) if z.shape[1] == 1 else torch.softmax(z, dim=-1)


def cls_dense_f32(t, w_nc, bias):
    return t[:, 0] @ w_nc.t() + (bias if bias is not None else 0)


def scale_add_act(x, scale=None, residual=None, act=None, act2=None):
    if act2 is not None:
        y = scale_add_act(x, scale, residual, act)
        return y, _r(R.act(y, _an(act2)), "saa")
    y = x
    if scale is not None:
        scale = _gate(scale)
        y = y * scale.reshape(scale.shape[0], *([1] * (x.dim() - 2)), scale.shape[-1])
    if residual is not None:
        y = y + residual
    return _r(R.act(y, _an(act)), "saa" if residual is None else "saa_res")


def mul(a, b, c, a_off=0, b_off=0):
    return _r(a[..., a_off:a_off + c] * b[..., b_off:b_off + c], "saa")


def radix_combine(x, scale, radix=2):
    B, H, W, RC = x.shape
    return _r((x * _gate(scale)[:, None, None, :]).reshape(B, H, W, radix, RC // radix).sum(3), "saa")


def window_attention(qkv, q_global, table, heads, ws, scale):
    from oracle import gcvit_ref
    B, Hp, Wp, CC = qkv.shape
    nq = 2 if q_global is not None else 3
    C = CC // nq
    hd = C // heads
    win = R.window_partition(qkv, ws).reshape(-1, ws * ws, nq, heads, hd).permute(2, 0, 3, 1, 4)
    B_ = win.shape[1]
    if q_global is not None:
        k, v = win[0], win[1]
        q = torch.repeat_interleave(q_global.reshape(B, ws * ws, C), B_ // B, dim=0).reshape(B_, ws * ws, heads, hd).permute(0, 2, 1, 3)
    else:
        q, k, v = win[0], win[1], win[2]
    o = gcvit_ref.window_attention_core(q, k, v, table, ws, scale)
    return _r(R.window_reverse(o.permute(0, 2, 1, 3).reshape(B_, ws * ws, C), ws, Hp, Wp, C), "attn")


def gcvit_attn_block(x, q_global, ln, qkv, proj, table, heads, ws, scale):
    y = dense(layernorm(x, ln[0], ln[1], float(ln[2])), qkv)
    return dense(window_attention(y, q_global, table, heads, ws, scale), proj, residual=x)


def mhsa(qkv, heads, scale):
    B, N, D3 = qkv.shape
    D = D3 // 3
    q, k, v = qkv.reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    attn = torch.softmax(scale * (q @ k.transpose(-1, -2)), dim=-1)
    return _r((attn @ v).permute(0, 2, 1, 3).reshape(B, N, D), "attn")


def vit_tokens(patches, cls, pos):
    B = patches.shape[0]
    return _r(torch.cat([cls.float().expand(B, 1, -1), patches], 1) + pos.float())


def to_device_nhwc8(x, device="cpu"):
    out = torch.zeros((*x.shape[:3], 8))
    out[..., :3] = x.to(torch.float16).float()
    return out


@contextlib.contextmanager
def patched(round_act=False):
    """swap vipcup_amd.ops' compute entry points for the CPU emulation (weights builders stay the product's)"""
    global ROUND_ACT
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    names = ["conv2d", "dense", "dense_split", "gap_ln_dense_f32", "head_prob", "mbconv_expand_dw", "mlp", "se_gate", "dwconv2d", "dwconv2d_se", "layernorm", "pool2d", "global_avgpool", "gap_dense_f32", "cls_dense_f32",
             "scale_add_act", "mul", "radix_combine", "window_attention", "gcvit_attn_block", "mhsa", "vit_tokens", "to_device_nhwc8"]
    saved = {n: getattr(ops, n) for n in names}
    old = ROUND_ACT
    ROUND_ACT = round_act
    try:
        for n in names:
            setattr(ops, n, globals()[n])
        yield
    finally:
        ROUND_ACT = old
        for n, f in saved.items():
            setattr(ops, n, f)
