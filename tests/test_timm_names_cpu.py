"""CPU: the timm (PyTorch) -> Keras variable map of the tfimm members (vip-cup-2022_amd/timm_names.py restates
/root/reference/models/tfimm/utils/timm.py:39-106,109-229): names both ways against the fragments extracted from the reference's
constructors (tests/golden/ref_varnames.json), layouts (OIHW -> HWIO, Dense transposed, layer-scale .gamma), and the round trip of a
whole member through a PyTorch-named .npz into zoo's checkpoint reader."""
import json
import os
import re

import numpy as np
import pytest
import torch

import vipcup_amd  # noqa: F401
from vipcup_amd import timm_names as T, zoo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_varnames.json")


def test_name_rule_examples():
    """hand-derived from the text of timm.py:58-104"""
    assert T.tf_to_timm_name("convnext_tiny/stem/0/kernel:0", 4, has_scope=True) == ("stem.0.weight", T.CONV2D)
    assert T.tf_to_timm_name("stages/1/downsample/1/kernel", 4) == ("stages.1.downsample.1.weight", T.CONV2D)
    assert T.tf_to_timm_name("stages/0/blocks/2/conv_dw/depthwise_kernel", 4) == ("stages.0.blocks.2.conv_dw.weight", T.CONV2D)
    assert T.tf_to_timm_name("blocks/3/attn/qkv/kernel", 2) == ("blocks.3.attn.qkv.weight", T.SIMPLE)
    assert T.tf_to_timm_name("blocks/3/norm1/gamma", 1) == ("blocks.3.norm1.weight", T.NO)
    assert T.tf_to_timm_name("blocks/3/norm1/beta", 1) == ("blocks.3.norm1.bias", T.NO)
    assert T.tf_to_timm_name("stages/0/blocks/0/gamma", 1) == ("stages.0.blocks.0.weight", T.NO)      # layer scale: PT ".gamma" renamed first (:121-135)
    assert T.tf_to_timm_name("cls_token", 3) == ("cls_token", T.NO)
    assert T.tf_to_timm_name("bn/moving_variance", 1) == ("bn.running_var", T.NO)
    assert T.tf_to_timm_name("a/x___b/kernel", 2) == ("a.b.weight", T.SIMPLE)                          # '$1___$2' -> $2
    assert T.tf_to_timm_name("layers_._0/remove/fc/kernel", 2) == ("layers.0.fc.weight", T.SIMPLE)


@pytest.mark.parametrize("member,family", [("vit_tiny_patch16_224", "tfimm_vit"), ("convnext_tiny_in22k", "tfimm_convnext")])
def test_timm_names_are_built_from_reference_fragments(member, family):
    """every PyTorch name the map produces is the reference's Keras name with '/' -> '.' and the five suffix renames - i.e. its path
    segments are the reference constructors' own `name=` fragments (ref_varnames.json, AST-extracted)"""
    golden = json.load(open(GOLDEN))[family]
    frags = {f["re"] for f in golden["fragments"]}
    rx = [re.compile(f + r"\Z") for f in frags]
    spec = zoo.MEMBERS[member]
    for name, v in spec.synth(spec.seed).items():
        pt, tr = T.tf_to_timm_name(name, v.dim())
        segs = pt.split(".")
        assert segs[-1] in ("weight", "bias", "cls_token", "pos_embed") or segs[-1] in name, (name, pt)
        path = "/".join(segs[:-1]) if segs[-1] in ("weight", "bias") else "/".join(segs)
        # greedy check: the Keras name and the PyTorch path agree segment by segment, and each level is a known fragment or an index
        assert path == "/".join(name.split("/")[:len(path.split("/"))]) or segs[-1] in ("cls_token", "pos_embed")
        for level in name.split("/")[:-1]:
            assert level.isdigit() or any(r.match(level) for r in rx) or any(r.match(level + "/0") for r in rx) or any(
                level in f for f in frags), (name, level)
        if v.dim() == 4:
            assert tr == T.CONV2D
        elif name.endswith("kernel"):
            assert tr == T.SIMPLE
        else:
            assert tr == T.NO


@pytest.mark.parametrize("member", ["vit_tiny_patch16_224", "convnext_tiny_in22k"])
def test_state_dict_round_trip_through_zoo(member, tmp_path):
    """Keras-named synthetic variables -> a PyTorch-named / PyTorch-laid-out state_dict (the INVERSE transposes, written here
    independently of timm_names) -> .npz -> zoo.read_checkpoint + match_variable_names: every variable bit-identical"""
    spec = zoo.MEMBERS[member]
    params = spec.synth(spec.seed)
    state = {}
    for name, v in params.items():
        a = v.numpy()
        segs = name.split("/")
        last = segs[-1]
        if last in ("kernel", "depthwise_kernel"):
            a = np.transpose(a, (3, 2, 0, 1)) if a.ndim == 4 else a.T          # HWIO -> OIHW / [in, out] -> [out, in]
            segs[-1] = "weight"
        elif last == "gamma":
            segs[-1] = "gamma" if "blocks" in segs and segs[-2].isdigit() else "weight"    # ConvNeXt layer scale keeps PyTorch's ".gamma"
        elif last == "beta":
            segs[-1] = "bias"
        state[".".join(segs)] = np.ascontiguousarray(a)
    state["some.bn.num_batches_tracked"] = np.zeros((), np.int64)
    assert T.looks_like_timm(state)
    np.savez(tmp_path / "timm.npz", **state)
    got = zoo.match_variable_names(spec, zoo.read_checkpoint(str(tmp_path / "timm.npz")))
    assert set(got) == set(params)
    for k, v in params.items():
        assert got[k].shape == v.shape and torch.equal(got[k], v), k
    # a missing PyTorch entry is an error that names it, as in the reference (:152-161)
    del state["head.fc.weight" if member.startswith("convnext") else "head.weight"]
    with pytest.raises(KeyError):
        T.from_timm_state_dict(state, {k: tuple(v.shape) for k, v in params.items()})
    assert not T.looks_like_timm(params)
