"""GCViT on the HIP operator set — host-side mirror of the reference package models/gcvit
(``GCViT`` models/gcvit.py:45-118, ``GCViTTiny`` :151-160, layers/{embedding,feature,level,block,attention}.py).

Layout notes: activations stay plain NHWC feature maps for the whole network; ``window_partition`` /
``window_reverse`` (layers/window.py:3-15) never materialise — the attention kernel addresses windows
inside the feature map, and the Dense layers around it are row-wise so they do not care.
"""
from typing import Dict

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen

# models/gcvit/models/gcvit.py:10-43
NAME2CONFIG = {
    "gcvit_xxtiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(2, 2, 6, 2), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    "gcvit_xtiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(3, 4, 6, 5), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    "gcvit_tiny": dict(window_size=(7, 7, 14, 7), dim=64, depths=(3, 4, 19, 5), num_heads=(2, 4, 8, 16), mlp_ratio=3.0),
    "gcvit_small": dict(window_size=(7, 7, 14, 7), dim=96, depths=(3, 4, 19, 5), num_heads=(3, 6, 12, 24), mlp_ratio=2.0,
                        layer_scale=1e-5),
    "gcvit_base": dict(window_size=(7, 7, 14, 7), dim=128, depths=(3, 4, 19, 5), num_heads=(4, 8, 16, 32), mlp_ratio=2.0,
                       layer_scale=1e-5),
}
KEEP_DIMS = [(False, False, False), (False, False), (True,), (True,)]  # models/gcvit.py:71
LN_EPS = 1e-5
PAD1 = (1, 1, 1, 1)


def synth_params(cfg: Dict, seed: int = 1002, classes: int = 1) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    dim = cfg["dim"]

    def conv_branch(name, c):
        g.dwconv(f"{name}/conv/0", 3, c, gain=1.0)
        g.dense(f"{name}/conv/2/fc/0", c, int(c * 0.25), bias=False, gain=2.0)
        g.dense(f"{name}/conv/2/fc/2", int(c * 0.25), c, bias=False)
        g.conv(f"{name}/conv/3", 1, 1, c, c, gain=0.5)

    def reduce_size(name, c, keep_dim):
        g.ln(f"{name}/norm1", c)
        conv_branch(name, c)
        co = c if keep_dim else 2 * c
        g.conv(f"{name}/reduction", 3, 3, c, co, gain=1.0)
        g.ln(f"{name}/norm2", co)

    g.conv("patch_embed/proj", 3, 3, 3, dim, bias=True, gain=1.0)
    reduce_size("patch_embed/conv_down", dim, True)
    c = dim
    n = len(cfg["depths"])
    for i in range(n):
        ws, heads = cfg["window_size"][i], cfg["num_heads"][i]
        for k, _ in enumerate(KEEP_DIMS[i]):
            conv_branch(f"levels/{i}/q_global_gen/to_q_global/{k}", c)
        for j in range(cfg["depths"][i]):
            b = f"levels/{i}/blocks/{j}"
            g.ln(f"{b}/norm1", c)
            g.dense(f"{b}/attn/qkv", c, c * (2 if j % 2 else 3))
            g.trunc_normal(f"{b}/attn/relative_position_bias_table", ((2 * ws - 1) ** 2, heads), 0.5)
            g.dense(f"{b}/attn/proj", c, c, gain=0.25)
            g.ln(f"{b}/norm2", c)
            g.dense(f"{b}/mlp/fc1", c, int(c * cfg["mlp_ratio"]), gain=2.0)
            g.dense(f"{b}/mlp/fc2", int(c * cfg["mlp_ratio"]), c, gain=0.25)
            if cfg.get("layer_scale") is not None:      # trained values (the 1e-5 init would mute both branches)
                g.raw(f"{b}/gamma1", g._u((c,), 0.5, 1.5))
                g.raw(f"{b}/gamma2", g._u((c,), 0.5, 1.5))
        if i < n - 1:
            reduce_size(f"levels/{i}/downsample", c, False)
            c *= 2
    g.ln("norm", c)
    g.dense("head", c, classes)
    return g.p


class _ConvBranch:
    """pad1 -> DWConv3x3 -> gelu -> SE -> Conv1x1 (+ residual), feature.py:90-98 / :130-138"""

    def __init__(self, p, name, dev):
        self.dw = ops.make_dw_weight(p[f"{name}/conv/0/depthwise_kernel"], None, dev)
        self.fc0 = ops.make_dense_weight(p[f"{name}/conv/2/fc/0/kernel"], None, dev)
        self.fc2 = ops.make_dense_weight(p[f"{name}/conv/2/fc/2/kernel"], None, dev)
        self.pw = ops.make_conv_weight(p[f"{name}/conv/3/kernel"], None, device=dev)

    def __call__(self, x):
        """returns x + branch(x)"""
        y, s = ops.dwconv2d_se(x, self.dw, None, 3, 1, PAD1, "gelu", self.fc0, self.fc2, "gelu", "sigmoid")
        return ops.conv2d(y, self.pw, residual=x, gate=s)    # y * s folded into the 1x1 conv's activation load


class _LN:
    def __init__(self, p, name, dev):
        self.g = p[f"{name}/gamma"].to(dev, torch.float32).contiguous()
        self.b = p[f"{name}/beta"].to(dev, torch.float32).contiguous()

    def __call__(self, x):
        return ops.layernorm(x, self.g, self.b, LN_EPS)


class _ReduceSize:
    """ReduceSize (feature.py:81-113)"""

    def __init__(self, p, name, dev):
        self.n1 = _LN(p, f"{name}/norm1", dev)
        self.n2 = _LN(p, f"{name}/norm2", dev)
        self.branch = _ConvBranch(p, name, dev)
        self.red = ops.make_conv_weight(p[f"{name}/reduction/kernel"], None, device=dev)

    def __call__(self, x, stride=2):
        x = self.branch(self.n1(x))
        return self.n2(ops.conv2d(x, self.red, stride=stride, pad=PAD1))


class _Block:
    """GCViTBlock (block.py:10-81) with WindowAttention (attention.py:7-83) and Mlp (feature.py:8-33)"""

    def __init__(self, p, name, ws, heads, global_query, dev):
        self.ws, self.heads, self.global_query = ws, heads, global_query
        self.n1 = _LN(p, f"{name}/norm1", dev)
        self.n2 = _LN(p, f"{name}/norm2", dev)
        self.qkv = ops.make_dense_weight(p[f"{name}/attn/qkv/kernel"], p[f"{name}/attn/qkv/bias"], dev)
        # layer scale (block.py:41-56,79-80): a per-channel gain on each residual branch = a gain on the rows of the
        # branch's last Dense, folded here
        g1 = p.get(f"{name}/gamma1")
        g2 = p.get(f"{name}/gamma2")

        def scaled(kernel, bias, g):
            return (kernel, bias) if g is None else (kernel * g[None, :], bias * g)
        self.proj = ops.make_dense_weight(*scaled(p[f"{name}/attn/proj/kernel"], p[f"{name}/attn/proj/bias"], g1), dev)
        self.table = p[f"{name}/attn/relative_position_bias_table"].to(dev, torch.float32).contiguous()
        self.fc1 = ops.make_dense_weight(p[f"{name}/mlp/fc1/kernel"], p[f"{name}/mlp/fc1/bias"], dev)
        self.fc2 = ops.make_dense_weight(*scaled(p[f"{name}/mlp/fc2/kernel"], p[f"{name}/mlp/fc2/bias"], g2), dev)

    def __call__(self, x, q_global):
        C = x.shape[-1]
        hd = C // self.heads
        # x + attn(norm1(x))  (gamma1 folded into proj, block.py:54-56,79): one launch at level 0, four elsewhere
        x = ops.gcvit_attn_block(x, q_global if self.global_query else None, (self.n1.g, self.n1.b, LN_EPS), self.qkv, self.proj,
                                 self.table, self.heads, self.ws, hd ** -0.5)
        return ops.mlp(x, self.fc1, self.fc2, act="gelu", residual=x, ln=(self.n2.g, self.n2.b, LN_EPS))   # x + mlp(norm2(x))  (:80)


@keras_predict
class GCViT:
    def __init__(self, params: Dict[str, torch.Tensor], window_size, dim, depths, num_heads, mlp_ratio=3.0,
                 layer_scale=None, classes: int = 1, device="cuda", first_strides: int = 2, head_act: str = "default"):
        p, dev = params, device
        self.first_strides = first_strides       # stride of the stem's conv_down reduction (embedding.py:8-16, feature.py:98; models/gcvit.py:47)
        self.head_act = head_act                 # models/gcvit.py:48,113
        self.cfg = dict(window_size=window_size, dim=dim, depths=depths, num_heads=num_heads, mlp_ratio=mlp_ratio)
        self.classes = classes
        self.stem = ops.make_conv_weight(p["patch_embed/proj/kernel"], p["patch_embed/proj/bias"], device=dev,
                                         pad_cin_to=8)
        self.stem_down = _ReduceSize(p, "patch_embed/conv_down", dev)
        self.levels = []
        n = len(depths)
        for i in range(n):
            lv = {
                "ws": window_size[i],
                "qgen": [(_ConvBranch(p, f"levels/{i}/q_global_gen/to_q_global/{k}", dev), kd)
                         for k, kd in enumerate(KEEP_DIMS[i])],
                "blocks": [_Block(p, f"levels/{i}/blocks/{j}", window_size[i], num_heads[i], bool(j % 2), dev)
                           for j in range(depths[i])],
                "down": _ReduceSize(p, f"levels/{i}/downsample", dev) if i < n - 1 else None,
            }
            self.levels.append(lv)
        self.norm = _LN(p, "norm", dev)
        self.head_w = p["head/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["head/bias"].to(dev, torch.float32)

    def _level(self, x, lv):
        """GCViTLevel.call (level.py:46-67)"""
        ws = lv["ws"]
        B, H, W, C = x.shape
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        if ph or pw:
            # FitWindow (feature.py:240-249): zero-pad BOTH sides to a multiple of the window, the odd pixel after - a 1 x 1
            # zero-pad "pooling" launch (vip_pool2d_nhwc_f16 with k = 1 copies in-range pixels and writes 0 elsewhere)
            x = ops.pool2d(x, 1, 1, (ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2), ops.POOL_MAX_ZEROPAD)
        qg = x
        for branch, keep_dim in lv["qgen"]:
            qg = branch(qg)
            if not keep_dim:
                qg = ops.pool2d(qg, 3, 2, PAD1, ops.POOL_MAX_ZEROPAD)
        qg = qg.reshape(B, ws * ws, C)
        for blk in lv["blocks"]:
            x = blk(x, qg)
        if ph or pw:
            x = ops.pool2d(x, 1, 1, (0, 0, 0, 0), ops.POOL_MAX_ZEROPAD, out_hw=(H, W))   # level.py:61 crops from the top-left corner (sic), as the reference does
        if lv["down"] is not None:
            x = lv["down"](x)
        return x

    def features(self, x, collect=None):
        """x: fp16 NHWC, RGB padded to 8 channels.  GCViT.forward_features (models/gcvit.py:98-105)."""
        assert x.shape[-1] == 8
        y = ops.conv2d(x, self.stem, stride=2, pad=PAD1)
        y = self.stem_down(y, stride=self.first_strides)
        if collect is not None:
            collect.append(y)
        for lv in self.levels:
            y = self._level(y, lv)
            if collect is not None:
                collect.append(y)
        return self.norm(y)

    def logits(self, x):
        return ops.gap_dense_f32(self.features(x), self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))


def GCViTTiny(params, classes=1, device="cuda", first_strides=2, head_act="default"):
    """models/gcvit.py:151-160"""
    return GCViT(params, **NAME2CONFIG["gcvit_tiny"], classes=classes, device=device, first_strides=first_strides, head_act=head_act)
