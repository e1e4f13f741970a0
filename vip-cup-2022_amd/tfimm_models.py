"""tfimm ViT and ConvNeXt on the HIP operator set — host-side mirrors of
models/tfimm/architectures/vit.py (``ViTConfig`` :35-118, ``ViTMultiHeadAttention`` :121-167, ``ViTBlock``
:170-227, ``ViT`` :290-467) and convnext.py (``ConvNeXtConfig`` :66-135, ``ConvNeXtBlock`` :147-229,
``ConvNeXtStage`` :232-296, ``ConvNeXt`` :299-438), with the shared layers of
models/tfimm/layers/transformers.py (``PatchEmbeddings`` :79-173, ``MLP`` :176-214).
Only the constructor arguments that change the inference graph are kept.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen

LN_EPS = 1e-6  # norm_layer "layer_norm_eps_1e-6" (vit.py:55, convnext.py:124)


class _LN:
    def __init__(self, p, name, dev):
        self.g = p[f"{name}/gamma"].to(dev, torch.float32).contiguous()
        self.b = p[f"{name}/beta"].to(dev, torch.float32).contiguous()

    def __call__(self, x):
        return ops.layernorm(x, self.g, self.b, LN_EPS)


# ------------------------------------------------------------------------------------------------
# ViT
# ------------------------------------------------------------------------------------------------
@dataclass
class ViTConfig:
    """vit.py:35-66 (inference-relevant fields)"""
    name: str
    input_size: Tuple[int, int] = (224, 224)
    patch_size: int = 16
    embed_dim: int = 768
    nb_blocks: int = 12
    nb_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    nb_classes: int = 1

    @property
    def grid_size(self):
        return (self.input_size[0] // self.patch_size, self.input_size[1] // self.patch_size)

    @property
    def nb_patches(self):
        return self.grid_size[0] * self.grid_size[1]


def variant(cfg, overrides: Dict):
    """config dataclass with the fields a checkpoint's model_config overrides (tfimm serialises the dataclass itself,
    models/serialization.py:75-76); unknown keys are an error, fields this graph does not read are ignored by the caller"""
    import dataclasses
    if not overrides:
        return cfg
    known = {f.name for f in dataclasses.fields(cfg)}
    bad = set(overrides) - known
    if bad:
        raise ValueError(f"{cfg.name}: unknown config fields {sorted(bad)}")
    return dataclasses.replace(cfg, **overrides)


VIT_CONFIGS = {
    "vit_tiny_patch16_224": ViTConfig("vit_tiny_patch16_224", embed_dim=192, nb_heads=3),    # vit.py:470-481
    "vit_small_patch16_224": ViTConfig("vit_small_patch16_224", embed_dim=384, nb_heads=6),  # vit.py:530-541
    "vit_base_patch16_224": ViTConfig("vit_base_patch16_224", embed_dim=768, nb_heads=12),   # vit.py:598-613
}


def vit_synth_params(cfg: ViTConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    D, ps = cfg.embed_dim, cfg.patch_size
    g.conv("patch_embed/proj", ps, ps, 3, D, bias=True, gain=1.0)
    g.raw("cls_token", g._n((1, 1, D), 0.5))
    g.raw("pos_embed", g._n((1, cfg.nb_patches + 1, D), 0.5))
    hid = int(D * cfg.mlp_ratio)
    for j in range(cfg.nb_blocks):
        b = f"blocks/{j}"
        g.ln(f"{b}/norm1", D)
        g.dense(f"{b}/attn/qkv", D, 3 * D, bias=cfg.qkv_bias)
        g.dense(f"{b}/attn/proj", D, D, gain=0.25)
        g.ln(f"{b}/norm2", D)
        g.dense(f"{b}/mlp/fc1", D, hid, gain=2.0)
        g.dense(f"{b}/mlp/fc2", hid, D, gain=0.25)
    g.ln("norm", D)
    g.dense("head", D, cfg.nb_classes)
    return g.p


@keras_predict
class ViT:
    def __init__(self, params: Dict[str, torch.Tensor], cfg: ViTConfig, device="cuda"):
        p, dev = params, device
        self.cfg = cfg
        D = cfg.embed_dim
        assert D // cfg.nb_heads == 64, "vip_mhsa_fwd_f16 implements head_dim 64"
        self.patch = ops.make_conv_weight(p["patch_embed/proj/kernel"], p["patch_embed/proj/bias"], device=dev, pad_cin_to=8)
        self.cls = p["cls_token"].reshape(D).to(dev, ops.act_dtype()).contiguous()      # stored like an activation (fp32 when strict)
        self.pos = p["pos_embed"].reshape(-1, D).to(dev, ops.act_dtype()).contiguous()
        self.blocks = []
        for j in range(cfg.nb_blocks):
            b = f"blocks/{j}"
            qkv_bias = p.get(f"{b}/attn/qkv/bias")
            self.blocks.append(dict(
                n1=_LN(p, f"{b}/norm1", dev), n2=_LN(p, f"{b}/norm2", dev),
                qkv=ops.make_dense_weight(p[f"{b}/attn/qkv/kernel"], qkv_bias, dev),
                proj=ops.make_dense_weight(p[f"{b}/attn/proj/kernel"], p[f"{b}/attn/proj/bias"], dev),
                fc1=ops.make_dense_weight(p[f"{b}/mlp/fc1/kernel"], p[f"{b}/mlp/fc1/bias"], dev),
                fc2=ops.make_dense_weight(p[f"{b}/mlp/fc2/kernel"], p[f"{b}/mlp/fc2/bias"], dev)))
        self.norm = _LN(p, "norm", dev)
        self.head_w = p["head/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["head/bias"].to(dev, torch.float32)

    def features(self, x, collect=None):
        """ViT.forward_features (vit.py:414-451) up to the final norm; returns [B, N, D] tokens."""
        cfg = self.cfg
        assert x.shape[-1] == 8 and tuple(x.shape[1:3]) == tuple(cfg.input_size), \
            "interpolate_input=False (vit.py:58): input must equal cfg.input_size"
        B = x.shape[0]
        ps = cfg.patch_size
        pe = ops.conv2d(x, self.patch, stride=ps)                       # PatchEmbeddings (transformers.py:131-139)
        t = ops.vit_tokens(pe.reshape(B, cfg.nb_patches, cfg.embed_dim), self.cls, self.pos)
        scale = (cfg.embed_dim // cfg.nb_heads) ** -0.5
        for blk in self.blocks:                                          # ViTBlock.call (vit.py:214-227)
            qkv = ops.dense(blk["n1"](t), blk["qkv"])
            att = ops.mhsa(qkv, cfg.nb_heads, scale)
            t = ops.dense(att, blk["proj"], residual=t)
            t = ops.mlp(t, blk["fc1"], blk["fc2"], act="gelu", residual=t, ln=(blk["n2"].g, blk["n2"].b, LN_EPS))
            if collect is not None:
                collect.append(t)
        return self.norm(t)

    def logits(self, x):
        t = self.features(x)                                             # head(norm(x)[:, 0]) (vit.py:441-461)
        return ops.cls_dense_f32(t, self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))


# ------------------------------------------------------------------------------------------------
# ConvNeXt
# ------------------------------------------------------------------------------------------------
@dataclass
class ConvNeXtConfig:
    """convnext.py:66-135 (inference-relevant fields)"""
    name: str
    patch_size: int = 4
    first_down: int = 1            # stem stride = first_down*2 (convnext.py:320-327): 2, NOT the stock 4
    embed_dim: Tuple = (96, 192, 384, 768)
    nb_blocks: Tuple = (3, 3, 9, 3)
    mlp_ratio: float = 4.0
    nb_classes: int = 1


CONVNEXT_CONFIGS = {
    "convnext_tiny_in22k": ConvNeXtConfig("convnext_tiny_in22k"),  # convnext.py:611-620
    # members of the earlier, larger ensembles (main.py:43-56 NAME2BS): same graph, wider / deeper
    "convnext_small_in22k": ConvNeXtConfig("convnext_small_in22k", nb_blocks=(3, 3, 27, 3)),                          # :623-632
    "convnext_base_in22k": ConvNeXtConfig("convnext_base_in22k", embed_dim=(128, 256, 512, 1024), nb_blocks=(3, 3, 27, 3)),    # :635-644
    "convnext_large_in22ft1k": ConvNeXtConfig("convnext_large_in22ft1k", embed_dim=(192, 384, 768, 1536), nb_blocks=(3, 3, 27, 3)),  # :518-527
    # the 384-pixel fine-tunes: the same graphs (the ensemble runs them at 200 x 200 anyway)            :575-596
    "convnext_base_384_in22ft1k": ConvNeXtConfig("convnext_base_384_in22ft1k", embed_dim=(128, 256, 512, 1024), nb_blocks=(3, 3, 27, 3)),
    "convnext_large_384_in22ft1k": ConvNeXtConfig("convnext_large_384_in22ft1k", embed_dim=(192, 384, 768, 1536), nb_blocks=(3, 3, 27, 3)),
}


def convnext_synth_params(cfg: ConvNeXtConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    g.conv("stem/0", cfg.patch_size, cfg.patch_size, 3, cfg.embed_dim[0], bias=True, gain=1.0)
    g.ln("stem/1", cfg.embed_dim[0])
    cprev = cfg.embed_dim[0]
    for j, (c, nb) in enumerate(zip(cfg.embed_dim, cfg.nb_blocks)):
        if j > 0:
            g.ln(f"stages/{j}/downsample/0", cprev)
            g.conv(f"stages/{j}/downsample/1", 2, 2, cprev, c, bias=True, gain=1.0)
        hid = int(cfg.mlp_ratio * c)
        for i in range(nb):
            b = f"stages/{j}/blocks/{i}"
            g.dwconv(f"{b}/conv_dw", 7, c, bias=True, gain=1.0)
            g.ln(f"{b}/norm", c)
            g.dense(f"{b}/mlp/fc1", c, hid, gain=2.0)
            g.dense(f"{b}/mlp/fc2", hid, c)
            g.raw(f"{b}/gamma", g._u((c,), 0.1, 0.4))   # layer scale (trained values; the 1e-6 init would mute the block)
        cprev = c
    g.ln("head/norm", cprev)
    g.dense("head/fc", cprev, cfg.nb_classes)
    return g.p


@keras_predict
class ConvNeXt:
    def __init__(self, params: Dict[str, torch.Tensor], cfg: ConvNeXtConfig, device="cuda"):
        p, dev = params, device
        self.cfg = cfg
        self.stem = ops.make_conv_weight(p["stem/0/kernel"], p["stem/0/bias"], device=dev, pad_cin_to=8)
        self.stem_norm = _LN(p, "stem/1", dev)
        self.stages = []
        for j, nb in enumerate(cfg.nb_blocks):
            st = {"down": None, "blocks": []}
            if j > 0:
                st["down"] = (_LN(p, f"stages/{j}/downsample/0", dev),
                              ops.make_conv_weight(p[f"stages/{j}/downsample/1/kernel"],
                                                   p[f"stages/{j}/downsample/1/bias"], device=dev))
            for i in range(nb):
                b = f"stages/{j}/blocks/{i}"
                gamma = p[f"{b}/gamma"]
                # x*gamma folded into fc2: (W x + b) * gamma = (W*gamma) x + b*gamma  (convnext.py:226)
                st["blocks"].append(dict(
                    dw=ops.make_dw_weight(p[f"{b}/conv_dw/depthwise_kernel"], None, dev),
                    dwb=p[f"{b}/conv_dw/bias"].to(dev, torch.float32).contiguous(),
                    norm=_LN(p, f"{b}/norm", dev),
                    fc1=ops.make_dense_weight(p[f"{b}/mlp/fc1/kernel"], p[f"{b}/mlp/fc1/bias"], dev),
                    fc2=ops.make_dense_weight(p[f"{b}/mlp/fc2/kernel"] * gamma, p[f"{b}/mlp/fc2/bias"] * gamma, dev)))
            self.stages.append(st)
        self.head_norm = _LN(p, "head/norm", dev)
        self.head_w = p["head/fc/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["head/fc/bias"].to(dev, torch.float32)

    def features(self, x, collect=None):
        """ConvNeXt.forward_features (convnext.py:376-406)"""
        assert x.shape[-1] == 8
        cfg = self.cfg
        y = self.stem_norm(ops.conv2d(x, self.stem, stride=cfg.first_down * 2))   # VALID, stride 2
        for st in self.stages:
            if st["down"] is not None:
                ln, cw = st["down"]
                y = ops.conv2d(ln(y), cw, stride=2)                               # 2x2/2 VALID (convnext.py:260-267)
            for blk in st["blocks"]:                                              # ConvNeXtBlock.call (:220-229)
                h = ops.dwconv2d(y, blk["dw"], blk["dwb"], 7, 1, (3, 3, 3, 3))
                n = blk["norm"]                                                   # LN fused into the MLP launch
                y = ops.mlp(h, blk["fc1"], blk["fc2"], act="gelu", residual=y, ln=(n.g, n.b, LN_EPS))
            if collect is not None:
                collect.append(y)
        return y

    def logits(self, x):
        f = self.features(x)                                                      # pool -> norm -> fc (:432-436)
        n = self.head_norm
        return ops.gap_ln_dense_f32(f, n.g, n.b, LN_EPS, self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))
