"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference tfimm ViT and ConvNeXt graphs
(models/tfimm/architectures/vit.py, convnext.py; layers/transformers.py).  PARITY UNPINNED (see
oracle/ops_ref.py header): no weights, tests or golden vectors ship for these models and TensorFlow is
unavailable.  Parameter names are the Keras variable paths of the reference layers.
"""
import torch

from . import ops_ref as R

LN_EPS = 1e-6

VIT = {  # name -> (embed_dim, nb_blocks, nb_heads, patch)   vit.py:470-481,530-541,598-613
    "vit_tiny_patch16_224": (192, 12, 3, 16),
    "vit_small_patch16_224": (384, 12, 6, 16),
    "vit_base_patch16_224": (768, 12, 12, 16),
}
CONVNEXT = {  # name -> (embed_dim, nb_blocks, patch_size, first_down)   convnext.py:611-620,66-135
    "convnext_tiny_in22k": ((96, 192, 384, 768), (3, 3, 9, 3), 4, 1),
    "convnext_small_in22k": ((96, 192, 384, 768), (3, 3, 27, 3), 4, 1),            # :623-632
    "convnext_base_in22k": ((128, 256, 512, 1024), (3, 3, 27, 3), 4, 1),           # :635-644
    "convnext_large_in22ft1k": ((192, 384, 768, 1536), (3, 3, 27, 3), 4, 1),       # :518-527
    "convnext_base_384_in22ft1k": ((128, 256, 512, 1024), (3, 3, 27, 3), 4, 1),    # :575-584
    "convnext_large_384_in22ft1k": ((192, 384, 768, 1536), (3, 3, 27, 3), 4, 1),   # :587-596
}


def _ln(p, name, x):
    return R.layernorm(x, p[f"{name}/gamma"], p[f"{name}/beta"], LN_EPS)


def _mlp(p, name, x):
    """MLP.call (layers/transformers.py:207-214)"""
    x = R.act(R.dense(x, p[f"{name}/fc1/kernel"], p[f"{name}/fc1/bias"]), "gelu")
    return R.dense(x, p[f"{name}/fc2/kernel"], p[f"{name}/fc2/bias"])


def vit_attention(p, name, x, heads):
    """ViTMultiHeadAttention.call (vit.py:148-167): scale applied to the logits"""
    B, N, D = x.shape
    hd = D // heads
    qkv = R.dense(x, p[f"{name}/qkv/kernel"], p.get(f"{name}/qkv/bias"))
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = torch.softmax((hd ** -0.5) * (q @ k.transpose(-1, -2)), dim=-1)
    o = (attn @ v).permute(0, 2, 1, 3).reshape(B, N, D)
    return R.dense(o, p[f"{name}/proj/kernel"], p[f"{name}/proj/bias"])


def interpolate_pos_embeddings(pos_embed, src_grid, tgt_grid, nb_tokens=1):
    """tfimm/layers/transformers.py:13-47: [1, N, D] -> [1, nb_tokens + gh*gw, D] via tf.image.resize(bicubic) of the patch part"""
    if tuple(src_grid) == tuple(tgt_grid):
        return pos_embed
    D = pos_embed.shape[-1]
    src = pos_embed[0, nb_tokens:].reshape(src_grid[0], src_grid[1], D)
    tgt = R.resize_bicubic(src, tgt_grid[0], tgt_grid[1]).reshape(1, tgt_grid[0] * tgt_grid[1], D)
    return torch.cat([pos_embed[:, :nb_tokens], tgt], dim=1)


def vit_forward_tokens(p, x, name, nb_blocks=None, collect=None, interpolate_input=False):
    """ViT.forward_features up to norm (vit.py:414-441); ``interpolate_input`` (vit.py:58,425-433): position embeddings resampled
    to the patch grid of the input"""
    D, nb, heads, ps = VIT[name]
    nb = nb_blocks or nb
    B = x.shape[0]
    t = R.conv2d(x, p["patch_embed/proj/kernel"], p["patch_embed/proj/bias"], ps)  # PatchEmbeddings, VALID
    grid = (t.shape[1], t.shape[2])
    t = t.reshape(B, -1, D)
    pos = p["pos_embed"]
    if interpolate_input:
        g0 = int(round((pos.shape[1] - 1) ** 0.5))
        pos = interpolate_pos_embeddings(pos, (g0, g0), grid)
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1) + pos
    for j in range(nb):
        b = f"blocks/{j}"
        t = t + vit_attention(p, f"{b}/attn", _ln(p, f"{b}/norm1", t), heads)   # ViTBlock.call (vit.py:214-227)
        t = t + _mlp(p, f"{b}/mlp", _ln(p, f"{b}/norm2", t))
        if collect is not None:
            collect.append(t)
    return _ln(p, "norm", t)


def vit_logits(p, x, name, nb_blocks=None, interpolate_input=False):
    """head(norm(x)[:, 0]) (vit.py:441-461)"""
    t = vit_forward_tokens(p, x, name, nb_blocks, interpolate_input=interpolate_input)
    return R.dense(t[:, 0], p["head/kernel"], p["head/bias"])


def convnext_features(p, x, name, nb_blocks=None, collect=None):
    """ConvNeXt.forward_features (convnext.py:376-406)"""
    dims, nbs, ps, first_down = CONVNEXT[name]
    nbs = nb_blocks or nbs
    x = R.conv2d(x, p["stem/0/kernel"], p["stem/0/bias"], first_down * 2)          # 4x4 stride 2 VALID (:320-327)
    x = _ln(p, "stem/1", x)
    for j, nb in enumerate(nbs):
        if j > 0:                                                                    # ConvNeXtStage.call (:283-296)
            x = _ln(p, f"stages/{j}/downsample/0", x)
            x = R.conv2d(x, p[f"stages/{j}/downsample/1/kernel"], p[f"stages/{j}/downsample/1/bias"], 2)
        for i in range(nb):                                                          # ConvNeXtBlock.call (:220-229)
            b = f"stages/{j}/blocks/{i}"
            h = R.dwconv2d(x, p[f"{b}/conv_dw/depthwise_kernel"], p[f"{b}/conv_dw/bias"], 1, (3, 3, 3, 3))
            h = _ln(p, f"{b}/norm", h)
            h = _mlp(p, f"{b}/mlp", h)
            x = x + h * p[f"{b}/gamma"]
        if collect is not None:
            collect.append(x)
    return x


def convnext_logits(p, x, name, nb_blocks=None):
    """pool -> head/norm -> head/fc (convnext.py:432-436)"""
    f = convnext_features(p, x, name, nb_blocks)
    return R.dense(_ln(p, "head/norm", R.global_avgpool(f)), p["head/fc/kernel"], p["head/fc/bias"])


def predict_logits(member, params, x):
    if member in VIT:
        return vit_logits(params, x, member)
    return convnext_logits(params, x, member)
