"""CPU: the architecture tables of the product graphs AND of the oracle restatements against the reference's own
tables, extracted as data (literals only, parsed with `ast`, nothing imported) by tools/extract_reference_configs.py
into tests/golden/ref_configs.json.  This pins WHAT is built - depths, widths, strides, expansion ratios, SE ratios,
window sizes, head counts, the manifest - to the reference; the numerics of each graph are the business of the GPU
parity tests."""
import inspect
import json
import math
import os

import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_configs.json")))


def _tup(v):
    return tuple(_tup(x) for x in v) if isinstance(v, (list, tuple)) else v


def test_manifest_and_batch_table():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble, zoo
    here = os.path.dirname(os.path.abspath(vipcup_amd.__file__))
    mine = json.load(open(os.path.join(here, "ckpts", "ckpts.json")))
    assert mine == GOLD["ckpts.json"]                                   # names, input sizes, fold index, ORDER
    for name, dim, _ in GOLD["ckpts.json"]:
        key = zoo.by_ckpt_name(name)
        assert key is not None, name
        assert [zoo.MEMBERS[key].input_hw] * 2 == dim
    # main.py:85  batch = 8 * NAME2BS.get(model_name, 16): no shipped member is in NAME2BS -> 128 for all of them
    assert not set(GOLD["main.py"]["NAME2BS"]) & {n for n, _, _ in GOLD["ckpts.json"]}
    assert ensemble.REF_BATCH == 8 * 16 and ensemble.NAME2BS == GOLD["main.py"]["NAME2BS"]
    assert ensemble.ref_batch("ResNetRS200-200x200") == 256 and ensemble.ref_batch("ResNetRS50-200x200") == 128
    # the earlier ensembles' members that are re-configurations of graphs built here carry the manifest's naming scheme
    assert not set(GOLD["main.py"]["NAME2BS"]) - {zoo.MEMBERS[k].ckpt_name for k in zoo.MEMBERS}     # all twelve are registered
    for key in ("resnet_rs200", "convnext_base_in22k", "convnext_large_in22ft1k", "gcvit_base", "resnest200", "eca_nfnet_l2", "resnet200d",
                "efficientnet_v2m", "efficientnet_v2l"):
        assert zoo.MEMBERS[key].ckpt_name in GOLD["main.py"]["NAME2BS"], key


def test_resnet_rs_block_args():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import resnet_rs
    from oracle import resnet_rs_ref
    ref = {int(d): [(s["input_filters"], s["num_repeats"]) for s in v] for d, v in GOLD["resnet_rs/block_args.py"]["BLOCK_ARGS"].items()}
    assert resnet_rs.BLOCK_ARGS == ref
    for d, v in resnet_rs_ref.BLOCK_ARGS.items():
        assert v == ref[d]
    assert {50, 101, 200} <= set(resnet_rs_ref.BLOCK_ARGS)


def test_gcvit_configs():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import gcvit
    from oracle import gcvit_ref
    ref = GOLD["gcvit/models/gcvit.py"]["NAME2CONFIG"]
    for table in (gcvit.NAME2CONFIG, gcvit_ref.NAME2CONFIG):
        for name, cfg in table.items():
            for k, v in cfg.items():
                assert _tup(ref[name][k]) == _tup(v), (name, k)
            assert ("layer_scale" in ref[name]) == ("layer_scale" in cfg), name
    assert set(gcvit.NAME2CONFIG) == set(gcvit_ref.NAME2CONFIG) == set(ref)


def test_efficientnet_tables():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import kecam_models as km
    from oracle import kecam_ref
    v2 = GOLD["efficientnet_v2.py"]
    v1 = GOLD["efficientnet_v1.py"]
    base2, t = v2["EfficientNetV2"]["defaults"], v2["EfficientNetV2T"]
    pops = {c["args"][0]: c["args"][1] for c in t["calls"] if c["func"] == "pop" and len(c["args"]) == 2}
    want_t = dict(expands=base2["expands"], out_channels=t["assigns"]["out_channels"], depthes=t["assigns"]["depthes"],
                  strides=base2["strides"], se_ratios=base2["se_ratios"], kernel_sizes=[base2["kernel_sizes"]] * 6,
                  first_conv_filter=pops["first_conv_filter"], output_conv_filter=pops["output_conv_filter"],
                  is_torch_mode=t["assigns"]["is_torch_mode"])
    base1 = v1["EfficientNetV1"]["defaults"]
    width, depth = [c for c in v1["EfficientNetV1B4"]["calls"] if c["func"] == "get_expanded_width_depth"][0]["args"]
    # get_expanded_width_depth (efficientnet_v1.py:9-18): the two base lists there are EfficientNetV1's own defaults
    want_b4 = dict(expands=base1["expands"], out_channels=[c * width for c in base1["out_channels"]],
                   depthes=[int(math.ceil(d * depth)) for d in base1["depthes"]], strides=base1["strides"],
                   se_ratios=base1["se_ratios"], kernel_sizes=base1["kernel_sizes"],
                   first_conv_filter=base1["first_conv_filter"] * width, output_conv_filter=base1["output_conv_filter"] * width,
                   is_torch_mode=base2["is_torch_mode"])
    for table in (km.EFFNET, kecam_ref.EFFNET):
        for name, want in (("EfficientNetV2T", want_t), ("EfficientNetV1B4", want_b4)):
            for k, v in want.items():
                got = table[name][k]
                if isinstance(v, list):
                    assert len(got) == len(v) and all(abs(a - b) < 1e-9 for a, b in zip(got, v)), (name, k, got, v)
                else:
                    assert got == pytest.approx(v), (name, k)
    assert base2["activation"] == "swish" and v1["EfficientNetV1B4"]["defaults"]["first_strides"] == 2
    for name in ("EfficientNetV2M", "EfficientNetV2L"):                 # efficientnet_v2.py:300-325
        a = v2[name]["assigns"]
        pops = {c["args"][0]: c["args"][1] for c in v2[name]["calls"] if c["func"] == "pop" and len(c["args"]) == 2}
        for table in (km.EFFNET, kecam_ref.EFFNET):
            t = table[name]
            for k in ("out_channels", "depthes", "expands", "strides", "se_ratios"):
                assert list(t[k]) == a[k], (name, k)
            assert t["kernel_sizes"] == [base2["kernel_sizes"]] * 7 and t["is_torch_mode"] == base2["is_torch_mode"]
            assert (t["first_conv_filter"], t["output_conv_filter"]) == (pops["first_conv_filter"], pops["output_conv_filter"])


def test_resnest_and_nfnet_tables():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import kecam_models as km
    from oracle import kecam_ref
    aot = GOLD["aotnet.py"]["AotNet"]["defaults"]
    rs = GOLD["resnest.py"]
    call = [c for c in rs["ResNest50"]["calls"] if c["func"] == "ResNest"][0]["kwargs"]
    assert _tup(km.RESNEST50["num_blocks"]) == _tup(call["num_blocks"]) and km.RESNEST50["stem_width"] == call["stem_width"]
    assert _tup(km.RESNEST50["out_channels"]) == _tup(aot["out_channels"]) and _tup(km.RESNEST50["strides"]) == _tup(aot["strides"])
    sig = inspect.signature(kecam_ref.resnest_features).parameters
    assert _tup(sig["num_blocks"].default) == _tup(call["num_blocks"]) and _tup(sig["out_channels"].default) == _tup(aot["out_channels"])
    assert _tup(sig["strides"].default) == _tup(aot["strides"])
    assert rs["ResNest50"]["defaults"]["groups"] == 2 and rs["ResNest"]["defaults"]["stem_type"] == "deep"   # radix 2, deep stem
    assert rs["ResNest"]["defaults"]["shortcut_type"] == "avg" and rs["ResNest"]["defaults"]["bn_after_attn"] is False

    nf = GOLD["nfnets.py"]
    l0, light, base = nf["ECA_NFNetL0"]["assigns"], nf["NormFreeNet_Light"]["defaults"], nf["NormFreeNet"]["defaults"]
    want = dict(num_blocks=l0["num_blocks"], out_channels=base["out_channels"], strides=base["strides"], stem_width=base["stem_width"],
                alpha=base["alpha"], channel_ratio=light["channel_ratio"], group_size=light["group_size"],
                num_features_factor=l0["num_features_factor"])
    for k, v in want.items():
        assert _tup(km.NFNET_L0[k]) == _tup(v), k
    sig = inspect.signature(kecam_ref.nfnet_features).parameters
    for k in ("num_blocks", "out_channels", "strides"):
        assert _tup(sig[k].default) == _tup(want[k]), k
    assert l0["attn_type"] == "eca" and nf["ECA_NFNetL0"]["defaults"]["activation"] == "swish"
    l2 = nf["ECA_NFNetL2"]["assigns"]                                   # nfnets.py:329-332: no factor given -> NormFreeNet's 2
    assert _tup(km.NFNET_L2["num_blocks"]) == _tup(l2["num_blocks"]) and "num_features_factor" not in l2
    assert km.NFNET_L2["num_features_factor"] == base["num_features_factor"] and l2["attn_type"] == "eca"
    rd = GOLD["resnet_deep.py"]                                          # ResNetD: deep stem, "avg" shortcut, no attention
    assert _tup(km.RESNET200D["num_blocks"]) == _tup(rd["ResNet200D"]["assigns"]["num_blocks"]) and km.RESNET200D["attn"] is None
    assert rd["ResNetD"]["defaults"]["stem_type"] == "deep" and rd["ResNetD"]["defaults"]["shortcut_type"] == "avg"
    assert aot["attn_types"] is None and aot["bn_after_attn"] is True and km.RESNET200D["stem_width"] == aot["stem_width"]
    c200 = [c for c in rs["ResNest200"]["calls"] if c["func"] == "ResNest"][0]["kwargs"]
    assert _tup(km.RESNEST200["num_blocks"]) == _tup(c200["num_blocks"]) and km.RESNEST200["stem_width"] == c200["stem_width"]
    assert light["torch_padding"] is True and light["gamma_in_act"] is False and light["use_zero_init_gain"] is False


def test_tfimm_configs():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import tfimm_models as tm
    from oracle import tfimm_ref
    for name in ("vit_tiny_patch16_224", "vit_small_patch16_224", "vit_base_patch16_224"):
        kw = [c for c in GOLD["vit.py"][name]["calls"] if c["func"] == "ViTConfig"][0]["kwargs"]
        cfg = tm.VIT_CONFIGS[name]
        assert (cfg.embed_dim, cfg.nb_blocks, cfg.nb_heads, cfg.patch_size) == (kw["embed_dim"], kw["nb_blocks"], kw["nb_heads"], kw["patch_size"])
        assert tfimm_ref.VIT[name] == (kw["embed_dim"], kw["nb_blocks"], kw["nb_heads"], kw["patch_size"])
    assert set(tm.CONVNEXT_CONFIGS) == set(GOLD["convnext.py"]) == set(tfimm_ref.CONVNEXT)
    for name in GOLD["convnext.py"]:
        kw = [c for c in GOLD["convnext.py"][name]["calls"] if c["func"] == "ConvNeXtConfig"][0]["kwargs"]
        cfg = tm.CONVNEXT_CONFIGS[name]
        assert kw["name"] == name and "first_down" not in kw and "patch_size" not in kw      # stride-2 4x4 stem (first_down = 1)
        assert kw.get("input_size", (224, 224)) in ((224, 224), [224, 224], (384, 384), [384, 384])
        assert _tup(cfg.embed_dim) == _tup(kw["embed_dim"]) and _tup(cfg.nb_blocks) == _tup(kw["nb_blocks"])
        assert (cfg.patch_size, cfg.first_down) == (4, 1)
        assert _tup(tfimm_ref.CONVNEXT[name]) == (_tup(kw["embed_dim"]), _tup(kw["nb_blocks"]), 4, 1)


def test_hornet_configs():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import hornet
    from oracle import hornet_ref
    hn = GOLD["hornet.py"]
    base = hn["HorNet"]["defaults"]
    for name, fn in (("hornet_tiny", "HorNetTiny"), ("hornet_small", "HorNetSmall"), ("hornet_base", "HorNetBase"),
                     ("hornet_large", "HorNetLarge")):
        want = dict(num_blocks=base["num_blocks"], embed_dim=hn[fn]["assigns"].get("embed_dim", base["embed_dim"]),
                    mlp_ratio=base["mlp_ratio"], gn_split=base["gn_split"], scale=base["scale"])
        assert "use_global_local_filter" not in hn[fn]["assigns"]          # the plain (depthwise 7x7) variants
        for table in (hornet.CONFIGS, hornet_ref.CONFIGS):
            for k, v in want.items():
                assert _tup(table[name][k]) == _tup(v), (name, k)
    assert hn["gnconv"]["defaults"]["dw_kernel_size"] == 7 and base["activation"] == "gelu" and base["layer_scale"] >= 0
    assert hornet.split_dims(128, 3) == [32, 64, 128]
