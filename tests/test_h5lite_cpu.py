"""The pure-Python HDF5 reader behind ``.h5`` checkpoint import (vipcup_amd/h5lite.py) against files written by h5py 3.3 / libhdf5
1.10.6 in the Keras weight / model layout (tools/make_h5_fixtures.py; expected arrays in the .npz next to each file)."""
import os

import numpy as np
import pytest

import vipcup_amd  # noqa: F401
from vipcup_amd import h5lite

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
FILES = ["keras_weights_small", "keras_model_many_layers", "keras_weights_latest_format"]


@pytest.mark.parametrize("name", FILES)
def test_keras_h5_weights_bit_exact(name):
    got = h5lite.load_keras_weights(os.path.join(GOLD, name + ".h5"))
    want = np.load(os.path.join(GOLD, name + ".npz"))
    assert list(got) == list(want.files)                       # Keras' own order: layer_names x weight_names
    for k in want.files:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert np.array_equal(got[k], want[k]), k


def test_tree_attributes_and_strings():
    f = h5lite.File(os.path.join(GOLD, "keras_model_many_layers.h5"))
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.8.0"          # variable-length strings (global heap)
    assert b'"class_name": "Functional"' in f.attrs["model_config"]
    mw = f["model_weights"]
    assert bytes(mw.attrs["backend"]) == b"tensorflow"                                               # fixed-length scalar string
    assert "layer_names" not in mw.attrs and "layer_names0" in mw.attrs                              # the chunked form Keras falls back to
    assert len(mw.keys()) == 43                                                                     # a split group B-tree
    ds = mw["big"]["big/kernel:0"]
    assert ds.shape == (3, 3, 40, 50) and ds.dtype == np.float32
    assert [k for k, _ in mw["block3_bn"].visit_datasets()] == ["block3_bn/beta:0", "block3_bn/gamma:0", "block3_bn/moving_mean:0",
                                                                "block3_bn/moving_variance:0"]
    with pytest.raises(KeyError):
        mw["no_such_layer"]


def test_not_hdf5_and_truncated(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p))
    raw = open(os.path.join(GOLD, "keras_weights_small.h5"), "rb").read()
    p.write_bytes(raw[:3000])
    with pytest.raises((h5lite.H5Error, ValueError)):
        h5lite.load_keras_weights(str(p))


def test_read_checkpoint_h5_equals_npz():
    """zoo.read_checkpoint: the .h5 and the .npz form of a checkpoint give the same {variable name: tensor}"""
    import torch
    from vipcup_amd import zoo
    a = zoo.read_checkpoint(os.path.join(GOLD, "keras_weights_small.h5"))
    b = zoo.read_checkpoint(os.path.join(GOLD, "keras_weights_small.npz"))
    assert list(a) == list(b)
    for k in a:
        assert isinstance(a[k], torch.Tensor) and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k])


H5PY_PYTHON = "/opt/conda/bin/python3.9"          # the one interpreter of this image that has h5py


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
def test_whole_member_checkpoint_through_libhdf5(tmp_path):
    """A complete member checkpoint (tfimm ViT-Tiny, 5.7 M parameters, 150+ variables with the reference's Keras names) written as a
    Keras .h5 by h5py and read back by h5lite: every variable bit-identical, and load_model accepts the file through the reference's
    ckpts/<member directory>/ckpt/<fold>.h5 layout (graph construction on the CPU only - no GPU call)."""
    import subprocess
    import torch
    from vipcup_amd import zoo
    key = "vit_tiny_patch16_224"
    params = zoo.build_params(key)
    ckpt_dir = tmp_path / "ckpts" / zoo.MEMBERS[key].ckpt_name / "ckpt"
    ckpt_dir.mkdir(parents=True)
    npz, h5 = str(tmp_path / "p.npz"), str(ckpt_dir / "0.h5")
    np.savez(npz, **{k: v.numpy() for k, v in params.items()})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([H5PY_PYTHON, os.path.join(root, "tools", "npz_to_keras_h5.py"), npz, h5], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = zoo.read_checkpoint(h5)
    assert set(got) == set(params)
    for k, v in params.items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    assert zoo.by_ckpt_name(os.path.basename(os.path.dirname(os.path.dirname(h5)))) == key      # what load_model keys the graph on


def test_damaged_files_are_reported_not_misread(tmp_path):
    """300 seeded corruptions of a fixture (byte flips, zeroed runs, truncations): the reader either returns arrays or raises H5Error -
    no other exception type, no hang (pointer cycles end in RecursionError -> H5Error), no absurd allocation."""
    import random
    raw = bytearray(open(os.path.join(GOLD, "keras_model_many_layers.h5"), "rb").read())
    rnd = random.Random(7)
    p = tmp_path / "m.h5"
    ok = bad = 0
    for i in range(300):
        b = bytearray(raw)
        kind = i % 3
        if kind == 0:
            for _ in range(rnd.randint(1, 8)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            o = rnd.randrange(len(b) - 64)
            b[o:o + rnd.randint(1, 64)] = bytes(rnd.randint(1, 64))[:0] or b"\0" * 8
        else:
            b = b[:rnd.randrange(100, len(b))]
        p.write_bytes(bytes(b))
        try:
            out = h5lite.load_keras_weights(str(p))
            assert all(isinstance(v, np.ndarray) for v in out.values())
            ok += 1
        except h5lite.H5Error:
            bad += 1
    assert ok + bad == 300 and bad > 50


def test_enclosing_model_scope_is_stripped():
    """variables exported from an enclosing Keras model (one common leading scope) still fit the member graph"""
    import torch
    from vipcup_amd import zoo
    key = "vit_tiny_patch16_224"
    spec = zoo.MEMBERS[key]
    params = zoo.build_params(key)
    scoped = {"vit_tiny_patch16_224/" + k: v for k, v in params.items()}
    got = zoo.match_variable_names(spec, scoped)
    assert set(got) == set(params) and all(torch.equal(got[k], params[k]) for k in params)
    assert zoo.match_variable_names(spec, params) is params                      # already fitting: untouched
    odd = dict(scoped)
    odd["other_scope/x"] = torch.zeros(1)
    assert zoo.match_variable_names(spec, odd) is odd                            # no single common scope: left for the constructor to report
