"""Graph variants from a checkpoint's Keras `model_config` (what tf.keras.models.load_model rebuilds the graph from, main.py:107):
extraction of first_strides / classes / head activation / input size from the JSON, the mapping onto the member constructors'
arguments, and the round trip through a full-model .h5 written by real libhdf5 (h5py under another interpreter of this image)."""
import json
import os
import subprocess

import numpy as np
import pytest

import vipcup_amd  # noqa: F401
from vipcup_amd import h5lite, zoo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY = "/opt/conda/bin/python3.9"


def functional(layers, name="m"):
    return {"class_name": "Functional", "config": {"name": name, "layers": layers}}


def layer(cls, **cfg):
    return {"class_name": cls, "config": cfg}


RS = functional([layer("InputLayer", batch_input_shape=[None, 200, 200, 3], name="input_1"),
                 layer("ZeroPadding2D", name="stem_pad_1", padding=[[1, 1], [1, 1]]),
                 layer("Conv2D", name="stem_conv_1", strides=[1, 1], filters=32),
                 layer("Conv2D", name="stem_conv_2", strides=[1, 1], filters=32),
                 layer("Conv2D", name="c2_block_0_se_reduce", strides=[1, 1], filters=64),
                 layer("Dense", name="predictions", units=2, activation="softmax")])


def test_functional_graph():
    info = zoo.variant_from_model_config(RS)
    assert info == {"input_hw": (200, 200), "stem_strides": 1, "stem_layer": "stem_conv_1", "n_layers": 7, "classes": 2, "head_act": "softmax"}
    assert zoo.variant_kwargs(zoo.MEMBERS["resnet_rs50"], info) == {"classes": 2, "first_strides": 1}
    # defaults are not passed on; a sigmoid on two classes (multi-label) is
    dflt = functional([layer("InputLayer", batch_input_shape=[None, 200, 200, 3]), layer("Conv2D", name="stem_conv", strides=[2, 2]),
                       layer("Dense", name="predictions", units=1, activation="sigmoid")])
    assert zoo.variant_kwargs(zoo.MEMBERS["efficientnet_v2t"], zoo.variant_from_model_config(dflt)) == {}
    ml = functional([layer("Conv2D", name="stem_conv", strides=2), layer("Dense", name="predictions", units=2, activation="sigmoid")])
    assert zoo.variant_kwargs(zoo.MEMBERS["eca_nfnet_l0"], zoo.variant_from_model_config(ml)) == {"classes": 2, "classifier_activation": "sigmoid"}
    lin = functional([layer("Conv2D", name="stem_1_conv", strides=[2, 2]), layer("Dense", name="predictions", units=1, activation=None)])
    assert zoo.variant_kwargs(zoo.MEMBERS["resnest50"], zoo.variant_from_model_config(lin)) == {"classifier_activation": "linear"}


def test_stem_stride_is_family_aware():
    """the serialised class of the NFNet stem is `nfnets>ScaledStandardizedConv2D` (nfnets.py:41), and HorNet's stem convolution
    runs at first_strides * 2 (hornet.py:144) - both with the class names Keras really writes."""
    nf = functional([layer("InputLayer", batch_input_shape=[None, 200, 200, 3]),
                     layer("ZeroPadding2D", name="stem_1_pad"),
                     layer("nfnets>ScaledStandardizedConv2D", name="stem_1_conv", strides=[1, 1], filters=16),
                     layer("nfnets>ScaledStandardizedConv2D", name="stem_2_conv", strides=[1, 1], filters=32),
                     layer("Dense", name="predictions", units=1, activation="sigmoid")])
    assert zoo.variant_kwargs(zoo.MEMBERS["eca_nfnet_l0"], zoo.variant_from_model_config(nf)) == {"first_strides": 1}
    nf2 = functional([layer("nfnets>ScaledStandardizedConv2D", name="stem_1_conv", strides=[2, 2]),
                      layer("Dense", name="predictions", units=1, activation="sigmoid")])
    assert zoo.variant_kwargs(zoo.MEMBERS["eca_nfnet_l0"], zoo.variant_from_model_config(nf2)) == {}
    # HorNet: default first_strides = 2 -> stem stride 4; first_strides = 1 -> stem stride 2
    hn = lambda st: functional([layer("InputLayer", batch_input_shape=[None, 200, 200, 3]),
                                layer("Conv2D", name="stem_conv", strides=[st, st], kernel_size=[4, 4]),
                                layer("DepthwiseConv2D", name="stack1_block1_gnconv_dw", strides=[1, 1]),
                                layer("Dense", name="predictions", units=1, activation="sigmoid")])
    assert zoo.variant_kwargs(zoo.MEMBERS["hornet_base"], zoo.variant_from_model_config(hn(4))) == {}
    assert zoo.variant_kwargs(zoo.MEMBERS["hornet_base"], zoo.variant_from_model_config(hn(2))) == {"first_strides": 1}
    with pytest.raises(ValueError):
        zoo.variant_kwargs(zoo.MEMBERS["hornet_base"], zoo.variant_from_model_config(hn(3)))
    # a depthwise layer is never the stem; a graph with layers and no stem convolution is refused, not defaulted
    nostem = functional([layer("InputLayer", batch_input_shape=[None, 200, 200, 3]), layer("DepthwiseConv2D", name="dw", strides=[1, 1]),
                         layer("Dense", name="predictions", units=1, activation="sigmoid")])
    with pytest.raises(ValueError):
        zoo.variant_kwargs(zoo.MEMBERS["resnet_rs50"], zoo.variant_from_model_config(nostem))


def test_wrong_input_size_is_refused():
    bad = functional([layer("InputLayer", batch_input_shape=[None, 256, 256, 3]), layer("Dense", name="predictions", units=1, activation="sigmoid")])
    with pytest.raises(ValueError):
        zoo.variant_kwargs(zoo.MEMBERS["resnet_rs50"], zoo.variant_from_model_config(bad))


def test_gcvit_custom_layers():
    """a functional export of the subclassed model lists its custom layers with their own configs (gcvit/layers/embedding.py:25-29)"""
    g = functional([layer("InputLayer", batch_input_shape=[None, 224, 224, 3]),
                    layer("gcvit>Stem", name="patch_embed", dim=64, first_strides=1),
                    layer("gcvit>GCViTLevel", name="levels/0", depth=3),
                    layer("Dense", name="head", units=2, activation="softmax")])
    info = zoo.variant_from_model_config(g)
    assert info["first_strides"] == 1 and info["classes"] == 2 and info["head_act"] == "softmax"
    assert zoo.variant_kwargs(zoo.MEMBERS["gcvit_tiny"], info) == {"classes": 2, "first_strides": 1}
    top = {"class_name": "gcvit>GCViT", "config": {"name": "gcvit_tiny", "num_classes": 1, "head_act": "sigmoid", "first_strides": 2}}
    assert zoo.variant_kwargs(zoo.MEMBERS["gcvit_tiny"], zoo.variant_from_model_config(top)) == {}


def test_tfimm_dataclass_config():
    """tfimm serialises the config dataclass itself (models/serialization.py:75-76)"""
    c = {"class_name": "Custom>ConvNeXt", "config": {"name": "convnext_tiny_in22k", "nb_classes": 2, "input_size": [200, 200], "patch_size": 4,
                                                      "first_down": 1, "embed_dim": [96, 192, 384, 768], "drop_rate": 0.0}}
    info = zoo.variant_from_model_config(c)
    assert info["classes"] == 2 and info["input_hw"] == (200, 200)
    kw = zoo.variant_kwargs(zoo.MEMBERS["convnext_tiny_in22k"], info)
    assert kw == {"nb_classes": 2, "patch_size": 4, "first_down": 1}
    from vipcup_amd import tfimm_models as tm
    v = tm.variant(tm.CONVNEXT_CONFIGS["convnext_tiny_in22k"], kw)
    assert v.nb_classes == 2 and v.first_down == 1
    with pytest.raises(ValueError):
        tm.variant(tm.CONVNEXT_CONFIGS["convnext_tiny_in22k"], {"no_such_field": 1})


@pytest.mark.skipif(not os.path.exists(H5PY), reason="no interpreter with h5py in this image")
def test_full_model_h5_round_trip(tmp_path):
    arrays = {"stem_conv_1/kernel": np.ones((3, 3, 3, 4), np.float32), "predictions/kernel": np.arange(8, dtype=np.float32).reshape(4, 2),
              "predictions/bias": np.zeros(2, np.float32)}
    np.savez(tmp_path / "p.npz", **arrays)
    (tmp_path / "c.json").write_text(json.dumps(RS))
    tool = os.path.join(ROOT, "tools", "npz_to_keras_h5.py")
    r = subprocess.run([H5PY, tool, str(tmp_path / "p.npz"), str(tmp_path / "full.h5"), str(tmp_path / "c.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([H5PY, tool, str(tmp_path / "p.npz"), str(tmp_path / "weights.h5")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert h5lite.load_keras_model_config(str(tmp_path / "full.h5")) == RS
    assert h5lite.load_keras_model_config(str(tmp_path / "weights.h5")) is None          # weight-only: constructor defaults
    got = h5lite.load_keras_weights(str(tmp_path / "full.h5"))
    assert sorted(got) == sorted(arrays) and all(np.array_equal(got[k], arrays[k]) for k in arrays)
    assert zoo.checkpoint_variant(zoo.MEMBERS["resnet_rs50"], str(tmp_path / "full.h5")) == {"classes": 2, "first_strides": 1}
    assert zoo.checkpoint_variant(zoo.MEMBERS["resnet_rs50"], str(tmp_path / "weights.h5")) == {}
    assert zoo.checkpoint_variant(zoo.MEMBERS["resnet_rs50"], str(tmp_path / "p.npz")) == {}
