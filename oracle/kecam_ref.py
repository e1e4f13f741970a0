"""TEST INFRASTRUCTURE — CPU fp32 restatement of the four keras_cv_attention_models ("kecam") members of
the shipped ensemble (ckpts/ckpts.json): ResNest50, EfficientNetV2T, EfficientNetV1B4, ECA_NFNetL0.
Follows models/keras_cv_attention_models/{common_layers.py, aotnet/aotnet.py, resnest/resnest.py,
efficientnet/efficientnet_v2.py, efficientnet/efficientnet_v1.py, nfnets/nfnets.py}.
PARITY UNPINNED (see oracle/ops_ref.py header).  Parameter names are the Keras layer names of the reference
(``stack1_block1_deep_1_conv/kernel`` ...).
"""
import math

import torch

from . import ops_ref as R


def make_divisible(vv, divisor=4, min_value=None, limit_round_down=0.9):
    """common_layers.py:398-406"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(vv + divisor / 2) // divisor * divisor)
    if new_v < limit_round_down * vv:
        new_v += divisor
    return new_v


def _bn(p, name, x, eps, act=None):
    """batchnorm_with_activation (common_layers.py:190-212)"""
    y = R.batchnorm(x, p[f"{name}bn/gamma"], p[f"{name}bn/beta"], p[f"{name}bn/moving_mean"],
                    p[f"{name}bn/moving_variance"], eps)
    return R.act(y, act)


def _conv(p, name, x, k=1, strides=1, padding="valid", torch_padding=True, groups=1, bias=False):
    """conv2d_no_bias (common_layers.py:230-248): torch mode = ZeroPadding2D(k//2) + VALID; else Keras SAME"""
    w = p[f"{name}conv/kernel"]
    b = p[f"{name}conv/bias"] if bias else None
    if padding == "same" and k // 2 > 0:
        if torch_padding:
            return R.conv2d(x, w, b, strides, (k // 2,) * 4, groups)
        return R.conv2d_same(x, w, b, strides, groups)
    return R.conv2d(x, w, b, strides, (0, 0, 0, 0), groups)


def se_module(p, name, x, se_ratio, divisor=8, limit_round_down=0.9, activation="relu"):
    """common_layers.py:311-332 (use_conv, use_bias)"""
    s = R.global_avgpool(x)[:, None, None, :]
    s = R.act(R.conv2d(s, p[f"{name}1_conv/kernel"], p[f"{name}1_conv/bias"]), activation)
    s = R.act(R.conv2d(s, p[f"{name}2_conv/kernel"], p[f"{name}2_conv/bias"]), "sigmoid")
    return x * s


def eca_kernel_size(filters, gamma=2.0, beta=1.0):
    tt = int((math.log(float(filters)) / math.log(2.0) + beta) / gamma)
    return max(tt if tt % 2 else tt + 1, 3)


def eca_module(p, name, x):
    """common_layers.py:335-353: GAP -> zero pad -> Conv1D(k, no bias) over the channel axis -> sigmoid"""
    C = x.shape[-1]
    k = eca_kernel_size(C)
    g = R.global_avgpool(x)                                    # [B, C]
    w = p[f"{name}conv1d/kernel"].reshape(1, 1, k)             # Keras Conv1D kernel [k, 1, 1]
    a = torch.nn.functional.conv1d(torch.nn.functional.pad(g, (k // 2, k // 2))[:, None, :], w)[:, 0, :]
    return x * torch.sigmoid(a)[:, None, None, :]


# ------------------------------------------------------------------------------------------------
# ResNest50 = AotNet(num_blocks=[3,4,6,3], stem "deep", attn "sa", bn_after_attn=False, shortcut "avg")
# ------------------------------------------------------------------------------------------------
def split_attention_conv2d(p, name, x, filters, strides, eps, groups=2, act="relu"):
    """resnest/resnest.py:27-66 (downsample_first=False)"""
    xs = torch.chunk(x, groups, dim=-1)
    logits = torch.cat([_conv(p, f"{name}1_g{i + 1}_", xs[i], 3, 1, "same") for i in range(groups)], dim=-1)
    logits = _bn(p, f"{name}1_", logits, eps, act)
    gap = sum(torch.chunk(logits, groups, dim=-1)).mean(dim=(1, 2), keepdim=True)
    a = R.conv2d(gap, p[f"{name}2_conv/kernel"], p[f"{name}2_conv/bias"])
    a = _bn(p, f"{name}2_", a, eps, act)
    a = R.conv2d(a, p[f"{name}3_conv/kernel"], p[f"{name}3_conv/bias"])
    B = a.shape[0]
    a = torch.softmax(a.reshape(B, 1, groups, -1), dim=2).reshape(B, 1, 1, -1)       # rsoftmax (:16-24)
    out = sum(torch.chunk(a * logits, groups, dim=-1))
    if strides > 1:
        out = R.avgpool_valid(out, 3, 2, (1, 1, 1, 1))                                # pad 1 + AvgPool 3x3/2 (:63-65)
    return out


def resnest_features(p, x, num_blocks=(3, 4, 6, 3), out_channels=(256, 512, 1024, 2048), strides=(1, 2, 2, 2),
                     stem_width=64, eps=1e-5, first_strides=2, collect=None, attn="sa"):
    """AotNet (aotnet/aotnet.py:284-377) specialised by ResNest (resnest.py:69-77)"""
    act = "relu"
    # deep_stem (:235-242) + stem_bn + pad/MaxPool (:326-330)
    x = _bn(p, "stem_1_", _conv(p, "stem_1_", x, 3, first_strides, "same"), eps, act)
    x = _bn(p, "stem_2_", _conv(p, "stem_2_", x, 3, 1, "same"), eps, act)
    x = _conv(p, "stem_3_", x, 3, 1, "same")
    x = _bn(p, "stem_", x, eps, act)
    x = R.maxpool_valid(x, 3, 2, (1, 1, 1, 1))
    for si, (nb, oc, st) in enumerate(zip(num_blocks, out_channels, strides)):
        for bi in range(nb):
            n = f"stack{si + 1}_block{bi + 1}_"
            s = st if bi == 0 else 1
            conv_shortcut = bi == 0 and (s != 1 or x.shape[-1] != oc)
            if conv_shortcut:                                      # conv_shortcut_branch, "avg" (:100-115)
                sc = R.avgpool_same(x, s, s) if s > 1 else x
                sc = _bn(p, f"{n}shortcut_", _conv(p, f"{n}shortcut_", sc, 1), eps)
            else:
                sc = x
            hid = int(oc * 0.25)
            d = _bn(p, f"{n}deep_1_", _conv(p, f"{n}deep_1_", x, 1), eps, act)       # deep_branch (:118-134)
            if attn == "sa":
                d = split_attention_conv2d(p, f"{n}deep_2_sa_", d, hid, s, eps)
            else:       # ResNetD (resnet_deep.py:13-16): attn_types None -> conv3x3 with the stride (aotnet.py:78-81) + BN + act (:89-91)
                d = _bn(p, f"{n}deep_2_", _conv(p, f"{n}deep_2_", d, 3, s, "same"), eps, act)
            d = _conv(p, f"{n}deep_3_", d, 1)
            d = _bn(p, f"{n}3_", d, eps)                                              # zero_gamma BN (:187)
            x = R.act(sc + d, act)
        if collect is not None:
            collect.append(x)
    return x


# ------------------------------------------------------------------------------------------------
# EfficientNetV2 / V1 (efficientnet_v2.py:47-193, efficientnet_v1.py:9-36)
# ------------------------------------------------------------------------------------------------
EFFNET = {
    "EfficientNetV2T": dict(expands=[1, 4, 4, 4, 6, 6], out_channels=[24, 40, 48, 104, 128, 208], depthes=[2, 4, 4, 6, 9, 14],
                            strides=[1, 2, 2, 2, 1, 2], se_ratios=[0, 0, 0, 0.25, 0.25, 0.25], kernel_sizes=[3] * 6,
                            first_conv_filter=24, output_conv_filter=1024, is_torch_mode=True),
    "EfficientNetV1B4": dict(expands=[1, 6, 6, 6, 6, 6, 6], out_channels=[ii * 1.4 for ii in [16, 24, 40, 80, 112, 192, 320]],
                             depthes=[int(math.ceil(ii * 1.8)) for ii in [1, 2, 2, 3, 3, 4, 1]],
                             strides=[1, 2, 2, 2, 1, 2, 1], se_ratios=[0.25] * 7, kernel_sizes=[3, 3, 5, 3, 5, 5, 3],
                             first_conv_filter=32 * 1.4, output_conv_filter=1280 * 1.4, is_torch_mode=False),
    # efficientnet_v2.py:300-325 (members of the earlier ensembles): TF-SAME padding / BN eps 1e-3 like every non-"T" variant
    "EfficientNetV2M": dict(expands=[1, 4, 4, 4, 6, 6, 6], out_channels=[24, 48, 80, 160, 176, 304, 512],
                            depthes=[3, 5, 5, 7, 14, 18, 5], strides=[1, 2, 2, 2, 1, 2, 1],
                            se_ratios=[0, 0, 0, 0.25, 0.25, 0.25, 0.25], kernel_sizes=[3] * 7, first_conv_filter=24,
                            output_conv_filter=1280, is_torch_mode=False),
    "EfficientNetV2L": dict(expands=[1, 4, 4, 4, 6, 6, 6], out_channels=[32, 64, 96, 192, 224, 384, 640],
                            depthes=[4, 7, 7, 10, 19, 25, 7], strides=[1, 2, 2, 2, 1, 2, 1],
                            se_ratios=[0, 0, 0, 0.25, 0.25, 0.25, 0.25], kernel_sizes=[3] * 7, first_conv_filter=32,
                            output_conv_filter=1280, is_torch_mode=False),
}


def _dw(p, name, x, k, stride, torch_mode):
    w = p[f"{name}MB_dw_/depthwise_kernel"]
    if torch_mode:
        return R.dwconv2d(x, w, None, stride, (k // 2,) * 4)
    pt, pb = R.same_pad(x.shape[1], k, stride)
    pl, pr = R.same_pad(x.shape[2], k, stride)
    return R.dwconv2d(x, w, None, stride, (pt, pb, pl, pr))


def inverted_residual_block(p, name, x, out, stride, expand, shortcut, k, se_ratio, is_fused, torch_mode, act="swish"):
    """efficientnet_v2.py:47-108"""
    eps = 1e-5 if torch_mode else 1e-3
    cin = x.shape[-1]
    hidden = make_divisible(cin * expand, 8)
    inp = x
    if is_fused and expand != 1:
        x = _bn(p, f"{name}sortcut_", _conv(p, f"{name}sortcut_", x, 3, stride, "same", torch_mode), eps, act)
    elif expand != 1:
        x = _bn(p, f"{name}sortcut_", _conv(p, f"{name}sortcut_", x, 1), eps, act)
    if not is_fused:
        x = _bn(p, f"{name}MB_dw_", _dw(p, name, x, k, stride, torch_mode), eps, act)
    if se_ratio > 0:
        x = se_module(p, f"{name}se_", x, se_ratio / expand, divisor=1, activation=act)
    if is_fused and expand == 1:
        x = _bn(p, f"{name}fu_", _conv(p, f"{name}fu_", x, 3, stride, "same", torch_mode), eps, act)
    else:
        x = _bn(p, f"{name}MB_pw_", _conv(p, f"{name}MB_pw_", x, 1), eps)
    return inp + x if shortcut else x


def effnet_features(p, x, name, first_strides=2, collect=None):
    """EfficientNetV2 (efficientnet_v2.py:111-193) up to the post conv"""
    c = EFFNET[name]
    tm = c["is_torch_mode"]
    eps = 1e-5 if tm else 1e-3
    act = "swish"
    x = _bn(p, "stem_", _conv(p, "stem_", x, 3, first_strides, "same", tm), eps, act)
    pre_out = make_divisible(c["first_conv_filter"], 8)
    for i, (e, oc, d, s, se, k) in enumerate(zip(c["expands"], c["out_channels"], c["depthes"], c["strides"],
                                                 c["se_ratios"], c["kernel_sizes"])):
        out = make_divisible(oc, 8)
        fused = se == 0
        for b in range(d):
            st = s if b == 0 else 1
            x = inverted_residual_block(p, f"stack_{i}_block{b}_", x, out, st, e, out == pre_out and st == 1, k, se,
                                        fused, tm, act)
            pre_out = out
        if collect is not None:
            collect.append(x)
    x = _bn(p, "post_", _conv(p, "post_", x, 1), eps, act)
    return x


# ------------------------------------------------------------------------------------------------
# ECA_NFNetL0 (nfnets/nfnets.py:42-320)
# ------------------------------------------------------------------------------------------------
SWISH_GAMMA = 1.7881293296813965  # NON_LINEAR_GAMMA["swish"] (nfnets.py:34)


def std_conv(p, name, x, k=1, strides=1, groups=1, gamma=SWISH_GAMMA, eps=1e-5):
    """ScaledStandardizedConv2D (nfnets.py:42-81): standardise over HWI per output channel, torch padding; the
    Keras Conv2D default use_bias=True applies (nfnets.py:99-108 passes no use_bias)."""
    w = p[f"{name}conv/kernel"]
    mean = w.mean(dim=(0, 1, 2), keepdim=True)
    var = w.var(dim=(0, 1, 2), unbiased=False, keepdim=True)
    fan_in = w.shape[0] * w.shape[1] * w.shape[2]
    scale = torch.rsqrt(torch.clamp(var * fan_in, min=eps)) * (p[f"{name}conv/gain"] * gamma)
    return R.conv2d(x, (w - mean) * scale, p[f"{name}conv/bias"], strides, (k // 2,) * 4, groups)


def nfnet_features(p, x, num_blocks=(1, 2, 6, 3), out_channels=(256, 512, 1536, 1536), strides=(1, 2, 2, 2),
                   stem_width=128, alpha=0.2, channel_ratio=0.25, group_size=64, num_features_factor=1.5,
                   first_strides=2, collect=None):
    """NormFreeNet_Light / ECA_NFNetL0: gamma_in_act=False -> conv_gamma = swish gamma, act_gamma = 1;
    use_zero_init_gain=False; attn_type="eca"."""
    act = "swish"
    # stem (:182-191): widths /8,/4,/2,1 ; strides first,1,1,2 ; activation after all but the last
    for i, (wd, s) in enumerate(zip((stem_width // 8, stem_width // 4, stem_width // 2, stem_width), (first_strides, 1, 1, 2))):
        x = std_conv(p, f"stem_{i + 1}_", x, 3, s)
        if i < 3:
            x = R.act(x, act)
    beta_list = [(1 + alpha ** 2 * ii) ** -0.5 for ii in range(max(num_blocks) + 1)]
    pre_beta = 1.0
    for si, (nb, oc, st) in enumerate(zip(num_blocks, out_channels, strides)):
        betas = beta_list[:nb + 1]
        betas[0] = pre_beta
        for bi in range(nb):
            n = f"stack{si + 1}_block{bi + 1}_"
            s = st if bi == 0 else 1
            hidden = int(oc * channel_ratio)
            groups = hidden // group_size
            preact = R.act(x, act) * betas[bi]                                        # block (:116-168)
            if s > 1 or x.shape[-1] != oc:
                sc = R.avgpool_same(preact, s, s) if s > 1 else preact
                sc = std_conv(p, f"{n}shortcut_", sc, 1)
            else:
                sc = x
            d = R.act(std_conv(p, f"{n}deep_1_", preact, 1), act)
            d = R.act(std_conv(p, f"{n}deep_2_", d, 3, s, groups), act)
            d = R.act(std_conv(p, f"{n}deep_3_", d, 3, 1, groups), act)
            d = std_conv(p, f"{n}deep_4_", d, 1)
            d = eca_module(p, f"{n}eca_", d) * 2.0                                    # attn_gain (:136,160-162)
            x = sc + d * alpha
        pre_beta = betas[-1]
        if collect is not None:
            collect.append(x)
    x = std_conv(p, "post_", x, 1)
    return R.act(x, act)


# ------------------------------------------------------------------------------------------------
ALIASES = {"resnest50": "ResNest50", "efficientnet_v2t": "EfficientNetV2T", "efficientnet_v1b4": "EfficientNetV1B4",
           "eca_nfnet_l0": "ECA_NFNetL0", "resnest200": "ResNest200", "eca_nfnet_l2": "ECA_NFNetL2",
           "efficientnet_v2m": "EfficientNetV2M", "efficientnet_v2l": "EfficientNetV2L", "resnet200d": "ResNet200D"}


def features(member, p, x, collect=None):
    member = ALIASES.get(member, member)
    if member == "ResNest50":
        return resnest_features(p, x, collect=collect)
    if member == "ResNet200D":                      # resnet_deep.py:34-36
        return resnest_features(p, x, num_blocks=(3, 24, 36, 3), attn=None, collect=collect)
    if member == "ResNest200":                      # resnest.py:84-85
        return resnest_features(p, x, num_blocks=(3, 24, 36, 3), stem_width=128, collect=collect)
    if member == "ECA_NFNetL2":                     # nfnets.py:329-332; num_features_factor: NormFreeNet's default 2
        return nfnet_features(p, x, num_blocks=(3, 6, 18, 9), num_features_factor=2, collect=collect)
    if member in EFFNET:
        return effnet_features(p, x, member, collect=collect)
    if member == "ECA_NFNetL0":
        return nfnet_features(p, x, collect=collect)
    raise KeyError(member)


def predict_logits(member, p, x):
    """GlobalAveragePooling2D -> Dense("predictions") (common_layers.py:278-283), pre-activation"""
    f = features(member, p, x)
    return R.dense(R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
