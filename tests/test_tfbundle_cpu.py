"""vipcup_amd/tfbundle.py - the SavedModel / TF-checkpoint variable reader (main.py:103-104,186-194: the reference falls back to
`ckpt/saved_model.pb`).  PARITY UNPINNED against TensorFlow itself (none here): the reader is held to an independent writer of the
published formats (tests/_tfbundle_writer.py), to the CRC32C known-answer vectors of RFC 3720, and to damage tests - it must refuse,
never guess."""
import os
import struct

import numpy as np
import pytest

import vipcup_amd  # noqa: F401
from tests import _tfbundle_writer as W
from vipcup_amd import tfbundle as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 and the classic check value
    assert T.crc32c(b"123456789") == 0xE3069283 == W._crc32c(b"123456789")
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283            # incremental form
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert T.crc_unmask(T.crc_mask(c)) == c and T.crc_mask(c) == W._mask(c)


def test_crc32c_vectorised_and_combine():
    g = np.random.default_rng(1)
    for n in (0, 1, 65535, 65536, 65537, 300001, (1 << 20) + 12345):
        data = g.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c_fast(data) == T.crc32c(data), n
    a, b = b"hello world, ", bytes(range(200)) * 3
    assert T.crc32c_combine(T.crc32c(a), T.crc32c(b), len(b)) == T.crc32c(a + b)
    assert T.crc32c_combine(T.crc32c(a), T.crc32c(b""), 0) == T.crc32c(a)


def test_large_tensor_checksum_is_verified(tmp_path):
    """a 6 MB variable: its checksum is verified too (vectorised CRC32C), a flipped byte in the middle is refused"""
    big = np.random.default_rng(2).standard_normal((1536, 1024)).astype(np.float32)
    prefix = str(tmp_path / "v" / "ckpt")
    # (the writer's bitwise CRC would take minutes on 6 MB: the reader's table-driven one, pinned by the known answers above, stands in)
    W_crc = W._crc32c
    W._crc32c = T.crc32c
    try:
        W.write_bundle(prefix, {"w": big}, block_size=4096)
    finally:
        W._crc32c = W_crc
    assert np.array_equal(T.load_tf_checkpoint(prefix)["w"], big)
    _flip(prefix + ".data-00000-of-00001", 3_000_001)
    with pytest.raises(T.BundleError, match="checksum"):
        T.load_tf_checkpoint(prefix)


def _variables(seed=0, n_layers=40):
    g = np.random.default_rng(seed)
    v = {}
    for i in range(n_layers):
        v[f"stack{i // 8}_block{i % 8}_conv/kernel"] = g.standard_normal((3, 3, 4 + i % 3, 8)).astype(np.float32)
        v[f"stack{i // 8}_block{i % 8}_conv/bias"] = g.standard_normal(8).astype(np.float32)
    v["head/kernel"] = g.standard_normal((64, 1)).astype(np.float32)
    v["head/bias"] = np.asarray([0.25], dtype=np.float32)
    return v


@pytest.mark.parametrize("block_size", [64, 512, 1 << 20])
def test_savedmodel_round_trip(tmp_path, block_size):
    """many data blocks (64-byte blocks: an index entry per key), a few, one; prefix-compressed keys across restart points"""
    v = _variables()
    cfg = {"class_name": "Functional", "config": {"name": "m", "layers": [
        {"class_name": "InputLayer", "config": {"batch_input_shape": [None, 200, 200, 3], "name": "input_1"}},
        {"class_name": "Conv2D", "config": {"name": "stem_conv", "strides": [1, 1], "filters": 24}},
        {"class_name": "Dense", "config": {"name": "head", "units": 2, "activation": "softmax"}}]}}
    d = tmp_path / "ckpt"
    W.write_savedmodel(str(d), v, cfg, block_size=block_size)
    for path in (str(d), str(d / "saved_model.pb")):                       # the reference hands over the .pb path's directory
        got = T.load_savedmodel_weights(path)
        assert set(got) == set(v)                                           # save counter / optimizer entries are not model variables
        for k in v:
            assert got[k].dtype == v[k].dtype and got[k].shape == v[k].shape and np.array_equal(got[k], v[k])
    assert T.load_savedmodel_config(str(d)) == cfg
    from vipcup_amd import zoo
    info = zoo.variant_from_model_config(T.load_savedmodel_config(str(d)))
    assert info == {"input_hw": (200, 200), "stem_strides": 1, "stem_layer": "stem_conv", "n_layers": 4, "classes": 2, "head_act": "softmax"}
    assert zoo.variant_kwargs(zoo.MEMBERS["resnet_rs50"], info) == {"classes": 2, "first_strides": 1}


def test_dtypes_scalars_and_name_based_checkpoints(tmp_path):
    t = {"a/x": np.arange(6, dtype=np.int32).reshape(2, 3), "a/y": np.asarray(2.5, dtype=np.float64), "b": np.asarray([True, False]),
         "h": np.asarray([1.5, -2.0], dtype=np.float16), "u": np.arange(5, dtype=np.uint8), "l": np.asarray([2 ** 40], dtype=np.int64)}
    prefix = str(tmp_path / "v" / "ckpt")
    W.write_bundle(prefix, t)                                               # no object graph: a TF1-style name-based checkpoint
    got = T.load_tf_checkpoint(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and np.array_equal(got[k], t[k]) and got[k].shape == t[k].shape
    b = T.Bundle(prefix)
    assert b.num_shards == 1 and b.object_graph_names() == {}


def test_string_scalar_and_object_graph(tmp_path):
    prefix = str(tmp_path / "v" / "ckpt")
    W.write_bundle(prefix, {"k1/.ATTRIBUTES/VARIABLE_VALUE": np.ones((2, 2), np.float32)}, [("k1/.ATTRIBUTES/VARIABLE_VALUE", "dense/kernel:0")],
                   string_entries={"note": b"hello \x00 world" * 40})
    b = T.Bundle(prefix)
    assert b.string_scalar("note") == b"hello \x00 world" * 40
    assert b.object_graph_names() == {"k1/.ATTRIBUTES/VARIABLE_VALUE": "dense/kernel:0"}
    assert list(T.load_tf_checkpoint(prefix)) == ["dense/kernel"]          # the ':0' of a variable name is dropped, as for .h5 files


def _flip(path, pos, xor=0x40):
    with open(path, "r+b") as f:
        f.seek(pos)
        b = f.read(1)
        f.seek(pos)
        f.write(bytes([b[0] ^ xor]))


def test_damage_is_refused(tmp_path):
    v = _variables(n_layers=6)
    d = tmp_path / "ckpt"
    W.write_savedmodel(str(d), v, None, block_size=128)
    index = str(d / "variables" / "variables.index")
    data = str(d / "variables" / "variables.data-00000-of-00001")
    size = os.path.getsize(index)
    good = open(index, "rb").read()
    # every single-byte flip in the index file is either caught (checksum, magic, structure) or leaves the decoded variables intact
    ref = T.load_savedmodel_weights(str(d))
    caught = 0
    for pos in range(0, size, 7):
        _flip(index, pos)
        try:
            got = T.load_savedmodel_weights(str(d))
            assert set(got) == set(ref) and all(np.array_equal(got[k], ref[k]) for k in ref), f"silent change at byte {pos}"
        except T.BundleError:
            caught += 1
        finally:
            open(index, "wb").write(good)
    assert caught >= (size // 7) * 0.9                                      # only footer padding bytes are free to change
    _flip(data, 40)                                                         # a weight byte: the tensor checksum
    with pytest.raises(T.BundleError, match="checksum"):
        T.load_savedmodel_weights(str(d))
    _flip(data, 40)
    os.rename(data, data + ".gone")
    with pytest.raises(T.BundleError, match="not found"):
        T.load_savedmodel_weights(str(d))
    os.rename(data + ".gone", data)
    open(index, "wb").write(good[:-8] + struct.pack("<Q", 0x1234))
    with pytest.raises(T.BundleError, match="magic"):
        T.load_savedmodel_weights(str(d))
    open(index, "wb").write(good)
    with pytest.raises(T.BundleError, match="saved_model.pb"):
        T.load_savedmodel_weights(str(tmp_path / "nowhere"))


def test_compressed_and_sliced_are_unsupported(tmp_path):
    p = str(tmp_path / "t.index")
    W.write_table(p, [(b"", W.pb_varint(1, 1)), (b"a", b"x")], compression_type=1)
    with pytest.raises(T.BundleError, match="compressed"):
        T.read_table(p)
    prefix = str(tmp_path / "s" / "ckpt")
    os.makedirs(os.path.dirname(prefix))
    entry = W.pb_varint(1, 1) + W.pb_bytes(2, W._shape_proto((4,))) + W.pb_varint(5, 16) + W.pb_bytes(7, b"\x0a\x00")
    W.write_table(prefix + ".index", [(b"", W.pb_varint(1, 1)), (b"w", entry)])
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(16))
    with pytest.raises(T.BundleError, match="sliced"):
        T.load_tf_checkpoint(prefix)


def test_zoo_reads_a_savedmodel_checkpoint(tmp_path):
    """zoo.read_checkpoint / checkpoint_variant on `ckpts/<member>/ckpt/saved_model.pb` (main.py:186-194): the same variables and
    the same graph variant as the .h5 form gives"""
    from vipcup_amd import zoo
    key = "vit_tiny_patch16_224"
    spec = zoo.MEMBERS[key]
    params = zoo.build_params(key, calibrated=False)
    d = tmp_path / "ckpts" / spec.ckpt_name / "ckpt"
    W.write_savedmodel(str(d), {k: v.numpy() for k, v in params.items()}, None, block_size=4096)
    got = zoo.read_checkpoint(str(d / "saved_model.pb"))
    assert set(got) == set(params) and all(np.array_equal(got[k].numpy(), params[k].numpy()) for k in params)
    assert zoo.checkpoint_variant(spec, str(d / "saved_model.pb")) == {}


def test_hand_assembled_bundle_from_the_published_formats():
    """tests/golden/tfbundle_hand/: written byte by byte by tools/make_tfbundle_fixture.py from the LevelDB table format and the
    tensor_bundle / tensor_shape / trackable_object_graph proto field numbers, with its own bit-at-a-time CRC32C, varints and block
    builder - no code shared with the reader or with tests/_tfbundle_writer.py.  Both restart intervals must give the expected variables."""
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfbundle_hand")
    exp = json.load(open(os.path.join(here, "expected.json")))
    ka = exp["known_answers"]
    assert ka["crc32c"]["123456789"] == 0xE3069283 and ka["crc32c"]["32 zero bytes"] == 0x8A9136AA       # RFC 3720 B.4
    assert ka["crc32c"]["32 0xff bytes"] == 0x62A8AB43 and ka["crc32c"]["0..31"] == 0x46DD794E
    assert T.crc32c(b"123456789") == ka["crc32c"]["123456789"] and T.crc32c(bytes(range(32))) == ka["crc32c"]["0..31"]
    assert T.crc_mask(T.crc32c(b"123456789")) == ka["masked(crc32c('123456789'))"]
    for ri in ("restart4", "restart16"):
        got = T.load_tf_checkpoint(os.path.join(here, ri, "variables"))
        assert set(got) == set(exp["variables"])                     # the save counter and the object graph are not model variables
        for name, e in exp["variables"].items():
            a = got[name]
            assert str(a.dtype) == e["dtype"] and list(a.shape) == e["shape"]
            assert np.array_equal(a.astype(np.float64).reshape(-1), np.array(e["values"]))


def test_unnamed_variables_are_refused_not_dropped():
    """an object graph whose attribute carries no full_name must not lose the weight silently (the constructor would name a
    missing variable much later); metric / optimizer / save-counter entries are bookkeeping and ARE left out"""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfbundle_hand")
    with pytest.raises(T.BundleError, match="no name recorded"):
        T.load_tf_checkpoint(os.path.join(here, "unnamed_variable", "variables"))
