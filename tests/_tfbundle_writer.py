"""Test infrastructure: an independent WRITER of the formats vipcup_amd/tfbundle.py reads - the LevelDB-style table file, the tensor
bundle index / data shard, the object-graph string tensor and keras_metadata.pb - built from the format descriptions, sharing no code
with the reader (its own varint / protobuf / CRC32C routines).  There is no TensorFlow here to write real fixtures with; this is what
the reader is exercised against (tests/test_tfbundle_cpu.py)."""
import json
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
DT = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("uint8"): 4, np.dtype("int64"): 9,
      np.dtype("bool"): 10, np.dtype("float16"): 19}


def _crc32c(data: bytes) -> int:            # bitwise (slow, obviously right)
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def _mask(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def vint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pb_varint(field: int, v: int) -> bytes:
    return vint(field << 3 | 0) + vint(v & ((1 << 64) - 1))


def pb_bytes(field: int, v: bytes) -> bytes:
    return vint(field << 3 | 2) + vint(len(v)) + v


def pb_fixed32(field: int, v: int) -> bytes:
    return vint(field << 3 | 5) + struct.pack("<I", v)


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += vint(shared) + vint(len(key) - shared) + vint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, pairs, block_size=512, restart_interval=16, compression_type=0):
    """pairs: sorted [(key bytes, value bytes)]"""
    out = bytearray()

    def emit(block: bytes):
        off = len(out)
        out.extend(block)
        out.append(compression_type)
        out.extend(struct.pack("<I", _mask(_crc32c(block + bytes([compression_type])))))
        return off, len(block)

    index = _BlockBuilder(1)
    cur = _BlockBuilder(restart_interval)
    for k, v in pairs:
        cur.add(k, v)
        if len(cur.buf) >= block_size:
            off, size = emit(cur.finish())
            index.add(cur.last, vint(off) + vint(size))
            cur = _BlockBuilder(restart_interval)
    if cur.count:
        off, size = emit(cur.finish())
        index.add(cur.last, vint(off) + vint(size))
    mi_off, mi_size = emit(_BlockBuilder(restart_interval).finish())          # empty metaindex block
    ix_off, ix_size = emit(index.finish())
    footer = vint(mi_off) + vint(mi_size) + vint(ix_off) + vint(ix_size)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


def _shape_proto(shape) -> bytes:
    return b"".join(pb_bytes(2, pb_varint(1, int(d))) for d in shape)


def write_bundle(prefix, tensors, object_graph=None, block_size=512, string_entries=None):
    """tensors: {checkpoint key: np.ndarray}; object_graph: [(checkpoint key, variable full_name)] -> the _CHECKPOINTABLE_OBJECT_GRAPH
    string tensor; string_entries: {key: bytes} further scalar string tensors"""
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    data = bytearray()
    entries = {}
    strings = dict(string_entries or {})
    if object_graph is not None:
        nodes = [pb_bytes(1, b"".join(pb_bytes(1, pb_varint(1, i + 1) + pb_bytes(2, f"layer-{i}".encode())) for i in range(len(object_graph))))]
        for key, full in object_graph:
            attr = pb_bytes(1, b"VARIABLE_VALUE") + pb_bytes(2, full.encode()) + pb_bytes(3, key.encode())
            nodes.append(pb_bytes(1, pb_bytes(2, attr)))
        strings["_CHECKPOINTABLE_OBJECT_GRAPH"] = b"".join(nodes)
    for key, arr in tensors.items():
        arr = np.asarray(arr)
        raw = arr.tobytes()                        # C order (np.ascontiguousarray would turn a scalar into shape (1,))
        entries[key] = (pb_varint(1, DT[arr.dtype]) + pb_bytes(2, _shape_proto(arr.shape)) + pb_varint(4, len(data)) + pb_varint(5, len(raw))
                        + pb_fixed32(6, _mask(_crc32c(raw))))
        data += raw
    for key, sval in strings.items():
        lens = vint(len(sval))
        raw = lens + struct.pack("<I", _mask(_crc32c(lens))) + sval
        entries[key] = pb_varint(1, 7) + pb_bytes(2, b"") + pb_varint(4, len(data)) + pb_varint(5, len(raw)) + pb_fixed32(6, 0)
        data += raw
    header = pb_varint(1, 1) + pb_varint(2, 0) + pb_bytes(3, pb_varint(1, 1))
    pairs = [(b"", header)] + sorted((k.encode(), v) for k, v in entries.items())
    write_table(prefix + ".index", pairs, block_size=block_size)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)


def write_savedmodel(directory, variables, model_config=None, block_size=512, extra_keys=True):
    """a Keras SavedModel directory holding `variables` {Keras variable name: array} under object-graph checkpoint keys"""
    os.makedirs(os.path.join(directory, "variables"), exist_ok=True)
    with open(os.path.join(directory, "saved_model.pb"), "wb") as f:
        f.write(b"\x08\x01")                     # the graph itself is not read
    tensors, graph = {}, []
    for i, (name, arr) in enumerate(variables.items()):
        key = f"layer_with_weights-{i // 2}/{name.rsplit('/', 1)[-1]}/.ATTRIBUTES/VARIABLE_VALUE"
        while key in tensors:
            key = "x" + key
        tensors[key] = arr
        graph.append((key, name))
    if extra_keys:                               # what a training checkpoint carries besides the model
        tensors["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] = np.asarray(3, dtype=np.int64)
        tensors["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.asarray(100, dtype=np.int64)
        graph += [("save_counter/.ATTRIBUTES/VARIABLE_VALUE", "save_counter"), ("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", "Adam/iter")]
    write_bundle(os.path.join(directory, "variables", "variables"), tensors, graph, block_size)
    if model_config is not None:
        md = json.dumps({"class_name": model_config["class_name"], "config": model_config["config"], "name": "model"}).encode()
        node = pb_varint(2, 0) + pb_bytes(3, b"root") + pb_bytes(4, b"_tf_keras_model") + pb_bytes(5, md)
        layer = pb_varint(2, 1) + pb_bytes(3, b"root.layer-0") + pb_bytes(4, b"_tf_keras_layer") + pb_bytes(5, b'{"name": "x"}')
        with open(os.path.join(directory, "keras_metadata.pb"), "wb") as f:
            f.write(pb_bytes(1, layer) + pb_bytes(1, node))
