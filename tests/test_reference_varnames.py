"""CPU: the Keras variable names this build's checkpoints use (the keys `main.py` / `zoo.load_model` read from .npz files, emitted by
`spec.synth()`) are pinned to the reference's constructors: every layer name must be a concatenation of name fragments that
tools/extract_reference_varnames.py found - with `ast`, nothing imported - in that family's source files (main.py:186-194 loads by
those names; tfimm/utils/timm.py:39-106 maps them), and every weight-carrying fragment on the members' constructor path must be used."""
import json
import os
import re

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_varnames.json")
# Keras' own variable names per layer class (tf.keras Conv2D / Dense / DepthwiseConv2D / BatchNormalization / LayerNormalization);
# custom variables (add_weight names) come from the extracted fragments
KERAS_VARS = {"kernel", "bias", "depthwise_kernel", "gamma", "beta", "moving_mean", "moving_variance"}
FAMILY = {"resnet_rs50": "resnet_rs", "gcvit_tiny": "gcvit", "convnext_tiny_in22k": "tfimm_convnext", "resnest50": "kecam_resnest",
          "efficientnet_v2t": "kecam_efficientnet", "efficientnet_v1b4": "kecam_efficientnet", "eca_nfnet_l0": "kecam_nfnet",
          "vit_small_patch16_224": "tfimm_vit", "vit_tiny_patch16_224": "tfimm_vit", "vit_base_patch16_224": "tfimm_vit"}
# weight-carrying fragments on the constructor path that no shipped configuration instantiates (reason each):
UNUSED_OK = {
    "resnet_rs": set(),
    "gcvit": {"gamma1", "gamma2"},                                   # layer scale: None for gcvit_tiny (models/gcvit.py:23-28, block.py:54-56)
    "tfimm_convnext": set(),
    "tfimm_vit": {"dist_token", "head_dist", "pre_logits/fc"},       # DeiT distillation token / head, representation layer (vit.py): not in ViT-Ti/S/B
    "kecam_resnest": set(),
    "kecam_efficientnet": {"1_dense", "2_dense"},                    # se_module(use_conv=False) branch (common_layers.py:324-330); EfficientNets use the conv form
    "kecam_nfnet": set(),
}


def _segment(name, frags):
    """can `name` be written as fragments joined directly or by '/'?  returns the list of fragments used, or None"""
    n = len(name)
    best = [None] * (n + 1)
    best[0] = []
    for i in range(n):
        if best[i] is None:
            continue
        j0 = i + 1 if (i > 0 and name[i] == "/") else i          # a '/' between two nested layers is free
        for src, rx in frags:
            m = rx.match(name, j0)
            if m and m.end() > j0 and best[m.end()] is None:
                best[m.end()] = best[i] + [src]
    return best[n]


@pytest.fixture(scope="module")
def golden():
    return json.load(open(GOLDEN))


@pytest.mark.parametrize("member", sorted(FAMILY))
def test_checkpoint_keys_are_built_from_reference_name_fragments(member, golden):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import zoo
    fam = golden[FAMILY[member]]
    frags = [(f["re"], re.compile(f["re"])) for f in {f["re"]: f for f in fam["fragments"]}.values()]
    custom_vars = {f["re"] for f in fam["fragments"] if f["callee"] == "add_weight"}
    spec = zoo.MEMBERS[member]
    keys = list(spec.synth(spec.seed))
    assert len(keys) > 50
    bad = []
    for k in keys:
        layer, _, var = k.rpartition("/")
        if layer == "" or (var not in KERAS_VARS and re.escape(var) not in custom_vars):
            # a custom variable that is addressed without a layer prefix (e.g. "cls_token")
            if _segment(k, frags) is None:
                bad.append(k)
            continue
        if _segment(layer, frags) is None:
            bad.append(k)
    assert not bad, f"{member}: {len(bad)} checkpoint keys are not made of reference name fragments, e.g. {bad[:8]}"


@pytest.mark.parametrize("family", sorted(UNUSED_OK))
def test_every_weight_carrying_fragment_on_the_path_is_used(family, golden):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import zoo
    fam = golden[family]
    frags = [(f["re"], re.compile(f["re"])) for f in {f["re"]: f for f in fam["fragments"]}.values()]
    used = set()
    for member, f in FAMILY.items():
        if f != family:
            continue
        spec = zoo.MEMBERS[member]
        for k in spec.synth(spec.seed):
            layer, _, var = k.rpartition("/")
            for cand in (layer, k):
                seg = _segment(cand, frags) if cand else None
                if seg:
                    used.update(seg)
            used.add(re.escape(var))
    want = {f["re"] for f in fam["fragments"] if f["on_path"] and f["callee"] in fam["weight_callees"]}
    missing = sorted(want - used - {re.escape(u) for u in UNUSED_OK[family]} - UNUSED_OK[family])
    assert not missing, f"{family}: weight-carrying reference layers with no counterpart in this build's checkpoints: {missing}"


def test_golden_file_is_what_the_extractor_writes():
    """the committed fixture equals a fresh AST pass (skipped where /root/reference is absent, e.g. on the GPU box)"""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "v.json")
        subprocess.run([sys.executable, os.path.join(root, "tools", "extract_reference_varnames.py"), "--ref", ref, "--out", out],
                       check=True, capture_output=True)
        assert json.load(open(out)) == json.load(open(GOLDEN))
