"""Hand-assembles a tiny TensorFlow checkpoint ("tensor bundle") byte by byte from the published formats - LevelDB table format
(doc/table_format.md: prefix-compressed entries, restart array, 1-byte type + masked CRC32C trailer, 48-byte footer with magic
0xdb4775248b80fb57), tensor_bundle.proto field numbers (BundleHeaderProto: 1 num_shards, 2 endianness, 3 version; BundleEntryProto:
1 dtype, 2 shape, 3 shard_id, 4 offset, 5 size, 6 crc32c fixed32), tensor_shape.proto (2 dim {1 size}), trackable_object_graph.proto
(1 nodes {2 attributes {1 name, 2 full_name, 3 checkpoint_key}}) and the string-tensor layout of tensor_bundle.cc (varint lengths, masked
CRC32C of the lengths, bytes) - and writes it under tests/golden/tfbundle_hand/ together with the expected variables.

It deliberately shares NO code with vip-cup-2022_amd/tfbundle.py (the reader under test) or tests/_tfbundle_writer.py (the other writer):
its CRC32C is the bit-at-a-time definition (reflected polynomial 0x82F63B78), its varints and blocks are assembled inline.  Two restart
intervals (4 and 16) exercise the prefix compression.  Run: python tools/make_tfbundle_fixture.py"""
import json
import os
import struct

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tfbundle_hand")


def crc_bits(data: bytes) -> int:                    # RFC 3720 B.4 (iSCSI CRC32C), one bit at a time
    reg = 0xFFFFFFFF
    for byte in data:
        reg ^= byte
        for _ in range(8):
            reg = (reg >> 1) ^ (0x82F63B78 if reg & 1 else 0)
    return reg ^ 0xFFFFFFFF


def masked(c: int) -> int:                           # crc32c.h: ((crc >> 15) | (crc << 17)) + 0xa282ead8
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def vint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def field(num: int, wire: int, payload) -> bytes:    # protobuf: key = (field << 3) | wire type
    key = vint((num << 3) | wire)
    if wire == 0:
        return key + vint(payload)
    if wire == 2:
        return key + vint(len(payload)) + payload
    if wire == 5:
        return key + struct.pack("<I", payload)
    raise ValueError(wire)


DT = {"float32": 1, "int64": 9, "float16": 19}
variables = {       # checkpoint key -> (graph name recorded in the object graph, array)
    "layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE": ("stem_conv/kernel:0", (np.arange(24, dtype=np.float32).reshape(1, 2, 3, 4) - 7.5) / 8),
    "layer_with_weights-0/bias/.ATTRIBUTES/VARIABLE_VALUE": ("stem_conv/bias:0", np.array([0.5, -1.25, 3.0, 1e-3], np.float32)),
    "layer_with_weights-1/gamma/.ATTRIBUTES/VARIABLE_VALUE": ("bn/gamma:0", np.array([1.0, 0.5, 2.0], np.float16)),
    "layer_with_weights-2/kernel/.ATTRIBUTES/VARIABLE_VALUE": ("predictions/kernel:0", np.linspace(-1, 1, 8, dtype=np.float32).reshape(4, 2)),
    "save_counter/.ATTRIBUTES/VARIABLE_VALUE": ("save_counter:0", np.array(3, np.int64)),
}

def build(unnamed=None):
    """(table(restart_interval) builder, data bytes); `unnamed`: a checkpoint key whose object-graph attribute gets NO full_name"""
    # ---- the data shard: tensors back to back; the object graph as a scalar string tensor ----------------------------------------------
    graph = b""
    for key, (full, _a) in variables.items():
        attr = field(1, 2, b"VARIABLE_VALUE") + (b"" if key == unnamed else field(2, 2, full.encode())) + field(3, 2, key.encode())
        graph += field(1, 2, field(2, 2, attr))          # one node per variable, one attribute each
    graph = field(1, 2, b"") + graph                     # the root node has no attributes
    lengths = vint(len(graph))
    string_tensor = lengths + struct.pack("<I", masked(crc_bits(lengths))) + graph

    data = bytearray()
    entries = {}


    def add(key: str, dtype: int, shape, raw: bytes, with_crc=True):
        dims = b"".join(field(2, 2, field(1, 0, d)) for d in shape)
        e = field(1, 0, dtype) + (field(2, 2, dims) if True else b"") + field(4, 0, len(data)) + field(5, 0, len(raw))
        if with_crc:
            e += field(6, 5, masked(crc_bits(raw)))
        entries[key] = e
        data.extend(raw)


    add("_CHECKPOINTABLE_OBJECT_GRAPH", 7, (), string_tensor, with_crc=False)
    for key, (_full, a) in variables.items():
        add(key, DT[str(a.dtype)], a.shape, a.astype(a.dtype.newbyteorder("<")).tobytes())
    header = field(1, 0, 1) + field(2, 0, 0) + field(3, 2, field(1, 0, 1))
    entries[""] = header


    # ---- the index: a LevelDB-format table ------------------------------------------------------------------------------------------------
    def block(pairs, restart_interval: int) -> bytes:
        body, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(pairs):
            if i % restart_interval == 0:
                restarts.append(len(body))
                shared = 0
            else:
                shared = 0
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            body += vint(shared) + vint(len(k) - shared) + vint(len(v)) + k[shared:] + v
            last = k
        for r in restarts:
            body += struct.pack("<I", r)
        body += struct.pack("<I", len(restarts))
        return bytes(body)


    def with_trailer(b: bytes) -> bytes:
        return b + b"\x00" + struct.pack("<I", masked(crc_bits(b + b"\x00")))


    def table(restart_interval: int) -> bytes:
        pairs = sorted((k.encode(), v) for k, v in entries.items())
        half = len(pairs) // 2                           # two data blocks, so that the index block has two handles
        out, handles = bytearray(), []
        for chunk in (pairs[:half], pairs[half:]):
            b = block(chunk, restart_interval)
            handles.append((chunk[-1][0], vint(len(out)) + vint(len(b))))
            out += with_trailer(b)
        meta = block([], 1)
        meta_handle = vint(len(out)) + vint(len(meta))
        out += with_trailer(meta)
        index = block(handles, 1)
        index_handle = vint(len(out)) + vint(len(index))
        out += with_trailer(index)
        footer = meta_handle + index_handle
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
        return bytes(out + footer)


    return table, bytes(data)


os.makedirs(OUT, exist_ok=True)
for name, ri, unnamed in (("restart4", 4, None), ("restart16", 16, None),
                          ("unnamed_variable", 16, "layer_with_weights-1/gamma/.ATTRIBUTES/VARIABLE_VALUE")):
    table, data = build(unnamed)
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "variables.index"), "wb").write(table(ri))
    open(os.path.join(d, "variables.data-00000-of-00001"), "wb").write(data)
expected = {full[:-2]: {"dtype": str(a.dtype), "shape": list(a.shape), "values": a.astype(np.float64).reshape(-1).tolist()}
            for key, (full, a) in variables.items() if not key.startswith("save_counter")}
vectors = {"crc32c": {"": crc_bits(b""), "123456789": crc_bits(b"123456789"), "32 zero bytes": crc_bits(bytes(32)),
                      "32 0xff bytes": crc_bits(b"\xff" * 32), "0..31": crc_bits(bytes(range(32)))},
           "masked(crc32c('123456789'))": masked(crc_bits(b"123456789"))}
json.dump({"variables": expected, "known_answers": vectors}, open(os.path.join(OUT, "expected.json"), "w"), indent=1)
print("wrote", OUT, len(data), "data bytes")
