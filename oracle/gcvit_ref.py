"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference GCViT graph (models/gcvit/).
PARITY UNPINNED (see oracle/ops_ref.py header): no weights, tests or golden vectors ship with the
reference for this model and TensorFlow / tensorflow_addons are unavailable.

Parameter names follow the Keras variable paths of the reference layers
(``levels/{i}/blocks/{j}/attn/qkv/kernel`` ...).
"""
import torch

from . import ops_ref as R

# models/gcvit/models/gcvit.py:10-43
NAME2CONFIG = {
    "gcvit_xxtiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(2, 2, 6, 2), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    "gcvit_xtiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(3, 4, 6, 5), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    "gcvit_tiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(3, 4, 19, 5), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    # models/gcvit.py:29-42: the two variants with a (trained) per-channel layer scale on both residual branches
    "gcvit_small": dict(window_size=(7, 7, 14, 7), dim=96, depths=(3, 4, 19, 5), num_heads=(3, 6, 12, 24), mlp_ratio=2.0,
                        layer_scale=1e-5),
    "gcvit_base": dict(window_size=(7, 7, 14, 7), dim=128, depths=(3, 4, 19, 5), num_heads=(4, 8, 16, 32), mlp_ratio=2.0,
                       layer_scale=1e-5),
}
KEEP_DIMS = [(False, False, False), (False, False), (True,), (True,)]  # models/gcvit.py:71
LN_EPS = 1e-5


def _ln(p, name, x):
    return R.layernorm(x, p[f"{name}/gamma"], p[f"{name}/beta"], LN_EPS)


def se(p, name, x):
    """SE (layers/feature.py:46-70): AdaptiveAvgPool(1) -> Dense(C/4, no bias) -> gelu -> Dense(C, no bias) -> sigmoid -> x*s"""
    s = R.global_avgpool(x)
    s = R.act(R.dense(s, p[f"{name}/fc/0/kernel"]), "gelu")
    s = R.act(R.dense(s, p[f"{name}/fc/2/kernel"]), "sigmoid")
    return x * s[:, None, None, :]


def _conv_branch(p, name, x):
    """pad1 -> DWConv3x3(no bias) -> gelu -> SE -> Conv1x1(no bias)   (feature.py:90-98,130-138)"""
    y = R.dwconv2d(x, p[f"{name}/conv/0/depthwise_kernel"], None, 1, (1, 1, 1, 1))
    y = R.act(y, "gelu")
    y = se(p, f"{name}/conv/2", y)
    return R.conv2d(y, p[f"{name}/conv/3/kernel"])


def reduce_size(p, name, x, first_strides=2):
    """ReduceSize.call (feature.py:104-113)"""
    x = _ln(p, f"{name}/norm1", x)
    x = x + _conv_branch(p, name, x)
    x = R.conv2d(x, p[f"{name}/reduction/kernel"], None, first_strides, (1, 1, 1, 1))
    return _ln(p, f"{name}/norm2", x)


def feat_extract(p, name, x, keep_dim):
    """FeatExtract.call (feature.py:144-153): zero-padded 3x3/2 max-pool unless keep_dim"""
    x = x + _conv_branch(p, name, x)
    if not keep_dim:
        x = R.maxpool_valid(x, 3, 2, (1, 1, 1, 1))
    return x


def fit_window(x, ws):
    """FitWindow.call (feature.py:240-249): pad both sides, the odd pixel goes after."""
    H, W = x.shape[1], x.shape[2]
    hp = (ws - H % ws) % ws
    wp = (ws - W % ws) % ws
    return R.zero_pad(x, (hp // 2, hp // 2 + hp % 2, wp // 2, wp // 2 + wp % 2))


def window_attention_core(q, k, v, table, ws, scale):
    """attention.py:69-79 — q,k,v [B_, heads, N, hd]; table [(2ws-1)^2, heads]"""
    q = q * scale
    attn = q @ k.transpose(-1, -2)
    idx = R.relative_position_index(ws).reshape(-1)
    bias = table[idx].reshape(ws * ws, ws * ws, -1).permute(2, 0, 1)
    attn = torch.softmax(attn + bias[None], dim=-1)
    return attn @ v


def window_attention(p, name, x, q_global, ws, heads):
    """WindowAttention.call (attention.py:52-83); x [B_, N, C]; q_global [B, ws, ws, C] or None"""
    B_, N, C = x.shape
    hd = C // heads
    nq = 2 if q_global is not None else 3
    qkv = R.dense(x, p[f"{name}/qkv/kernel"], p[f"{name}/qkv/bias"])
    qkv = qkv.reshape(B_, N, nq, heads, hd).permute(2, 0, 3, 1, 4)
    if q_global is not None:
        k, v = qkv[0], qkv[1]
        B = q_global.shape[0]
        qg = torch.repeat_interleave(q_global, B_ // B, dim=0)
        q = qg.reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    else:
        q, k, v = qkv[0], qkv[1], qkv[2]
    o = window_attention_core(q, k, v, p[f"{name}/relative_position_bias_table"], ws, hd ** -0.5)
    o = o.permute(0, 2, 1, 3).reshape(B_, N, C)
    return R.dense(o, p[f"{name}/proj/kernel"], p[f"{name}/proj/bias"])


def mlp(p, name, x):
    """Mlp.call (feature.py:26-33)"""
    x = R.act(R.dense(x, p[f"{name}/fc1/kernel"], p[f"{name}/fc1/bias"]), "gelu")
    return R.dense(x, p[f"{name}/fc2/kernel"], p[f"{name}/fc2/bias"])


def block(p, name, x, q_global, ws, heads):
    """GCViTBlock.call (block.py:60-81); gamma1 / gamma2 exist only when the config has a layer_scale (:41-56)."""
    B, H, W, C = x.shape
    y = _ln(p, f"{name}/norm1", x)
    y = R.window_partition(y, ws).reshape(-1, ws * ws, C)
    y = window_attention(p, f"{name}/attn", y, q_global, ws, heads)
    y = R.window_reverse(y, ws, H, W, C)
    g1 = p.get(f"{name}/gamma1")
    g2 = p.get(f"{name}/gamma2")
    x = x + (y if g1 is None else y * g1)                                  # :79
    m = mlp(p, f"{name}/mlp", _ln(p, f"{name}/norm2", x))
    return x + (m if g2 is None else g2 * m)                               # :80


def level(p, name, x, depth, heads, ws, keep_dims, downsample):
    """GCViTLevel.call (level.py:46-67)"""
    H, W = x.shape[1], x.shape[2]
    x = fit_window(x, ws)
    qg = x
    for i, kd in enumerate(keep_dims):
        qg = feat_extract(p, f"{name}/q_global_gen/to_q_global/{i}", qg, kd)
    for i in range(depth):
        x = block(p, f"{name}/blocks/{i}", x, qg if i % 2 else None, ws, heads)
    x = x[:, :H, :W, :]
    if downsample:
        x = reduce_size(p, f"{name}/downsample", x)
    return x


def forward_features(p, x, cfg, collect=None, first_strides=2):
    """GCViT.forward_features (models/gcvit.py:98-105): Stem -> levels -> LN"""
    # Stem (layers/embedding.py:8-23): ZeroPad(1) -> Conv3x3/2 (bias) -> ReduceSize(keep_dim, first_strides): the constructor's
    # `first_strides` (models/gcvit.py:47,70) is the stride of conv_down's reduction conv (feature.py:98), the proj conv is always /2
    x = R.conv2d(x, p["patch_embed/proj/kernel"], p["patch_embed/proj/bias"], 2, (1, 1, 1, 1))
    x = reduce_size(p, "patch_embed/conv_down", x, first_strides)
    if collect is not None:
        collect.append(x)
    n = len(cfg["depths"])
    for i in range(n):
        x = level(p, f"levels/{i}", x, cfg["depths"][i], cfg["num_heads"][i], cfg["window_size"][i], KEEP_DIMS[i],
                  i < n - 1)
        if collect is not None:
            collect.append(x)
    return _ln(p, "norm", x)


def forward_logits(p, x, cfg, first_strides=2):
    """forward_head (models/gcvit.py:107-113): GAP -> Dense (pre-activation)"""
    f = forward_features(p, x, cfg, first_strides=first_strides)
    return R.dense(R.global_avgpool(f), p["head/kernel"], p["head/bias"])


def predict_logits(member, params, x):
    """Uniform entry used by tests / bench: member name -> logits."""
    return forward_logits(params, x, NAME2CONFIG[member])
