"""TEST INFRASTRUCTURE - fp32 CPU restatement of kecam HorNet (models/keras_cv_attention_models/hornet/hornet.py), the
recursive-gated-convolution member of the reference's earlier ensembles.  Only tests/ may import this.  PARITY
UNPINNED: no golden vectors, TensorFlow not importable here (the architecture table is pinned by
tests/test_reference_configs.py)."""
import torch

from . import ops_ref as R

LN_EPS = 1e-5                     # common_layers.py:8
CONFIGS = {                       # hornet.py:127-135 defaults; :181, :186-187, :196-197, :206-207
    "hornet_tiny": dict(num_blocks=(2, 3, 18, 2), embed_dim=64, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_small": dict(num_blocks=(2, 3, 18, 2), embed_dim=96, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_base": dict(num_blocks=(2, 3, 18, 2), embed_dim=128, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_large": dict(num_blocks=(2, 3, 18, 2), embed_dim=192, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
}


def _ln(p, name, x):
    return R.layernorm(x, p[f"{name}/gamma"], p[f"{name}/beta"], LN_EPS)


def _conv(p, name, x, stride=1):
    return R.conv2d(x, p[f"{name}/kernel"], p[f"{name}/bias"], stride, (0, 0, 0, 0), 1)


def gnconv(p, name, x, gn_split, scale):
    """hornet.py:84-107"""
    c = x.shape[-1]
    nn = _conv(p, f"{name}pre_conv", x)                                     # C -> 2C
    dims = [c // (2 ** i) for i in range(gn_split)][::-1]
    pw_first, dw_list = nn[..., :dims[0]], nn[..., dims[0]:]
    dw_list = R.dwconv2d(dw_list, p[f"{name}list_dw_conv/depthwise_kernel"], p[f"{name}list_dw_conv/bias"], 1, (3, 3, 3, 3))
    dw_list = dw_list * scale
    parts = torch.split(dw_list, dims, dim=-1)
    nn = pw_first * parts[0]
    for i, dw in enumerate(parts[1:], start=1):
        nn = _conv(p, f"{name}pw{i}_conv", nn) * dw
    return _conv(p, f"{name}output_conv", nn)


def block(p, name, x, mlp_ratio, gn_split, scale):
    """hornet.py:110-124 (layer_scale >= 0: ChannelAffine gammas present)"""
    a = gnconv(p, f"{name}gnconv_", _ln(p, f"{name}attn_ln", x), gn_split, scale)
    x = x + a * p[f"{name}1_gamma/weight"]
    m = _ln(p, f"{name}mlp_ln", x)
    m = R.act(R.dense(m, p[f"{name}mlp_Dense_0/kernel"], p[f"{name}mlp_Dense_0/bias"]), "gelu")
    m = R.dense(m, p[f"{name}mlp_Dense_1/kernel"], p[f"{name}mlp_Dense_1/bias"])
    return x + m * p[f"{name}2_gamma/weight"]


def forward_features(p, x, cfg, first_strides=2, collect=None):
    """hornet.py:142-164; x [B,H,W,3] fp32"""
    x = _ln(p, "stem_ln", _conv(p, "stem_conv", x, first_strides * 2))
    for si, nb in enumerate(cfg["num_blocks"]):
        st = f"stack{si + 1}_"
        if si > 0:
            x = _conv(p, f"{st}conv", _ln(p, f"{st}ln", x), 2)
        for bi in range(nb):
            x = block(p, f"{st}block{bi + 1}_", x, cfg["mlp_ratio"], cfg["gn_split"][si], cfg["scale"])
        if collect is not None:
            collect.append(x)
    return x


def forward_logits(p, x, cfg, first_strides=2):
    """avg_pool -> pre_output_ln -> Dense, pre-activation (hornet.py:166-171)"""
    v = _ln(p, "pre_output_ln", R.global_avgpool(forward_features(p, x, cfg, first_strides)))
    return R.dense(v, p["predictions/kernel"], p["predictions/bias"])


def predict_logits(member, params, x):
    return forward_logits(params, x, CONFIGS[member])
