"""Read-only TensorFlow checkpoint ("tensor bundle") and Keras SavedModel variable reader in pure Python - the second checkpoint form
the reference accepts: ``ckpts/<name>/ckpt/saved_model.pb`` with ``variables/variables.{index,data-00000-of-00001}`` next to it
(main.py:103-104,186-194: ``tf.keras.models.load_model`` on the directory).

PARITY UNPINNED: no TensorFlow exists in this environment and the reference ships no SavedModel, so this reader has never seen a file
written by TensorFlow itself.  It is written from the published formats - the LevelDB table format that ``tensorflow/core/lib/io/table*``
port (footer with two block handles + magic 0xdb4775248b80fb57, prefix-compressed entries with restart points, 1-byte compression type +
masked CRC32C trailer per block), ``tensor_bundle.proto`` (BundleHeaderProto, BundleEntryProto), the string-tensor layout of
``tensor_bundle.cc`` and ``trackable_object_graph.proto`` - and is exercised against an independent writer of the same formats
(tests/_tfbundle_writer.py) plus damage tests.  It refuses what it does not implement (snappy-compressed blocks, sliced / partitioned
variables, big-endian bundles, multi-shard files it cannot find) instead of guessing, and checks every block CRC and every tensor CRC.

    load_savedmodel_weights(path)  ->  {Keras variable name: np.ndarray}      path = the SavedModel directory or its saved_model.pb
    load_savedmodel_config(path)   ->  the Keras model config (dict) from keras_metadata.pb, or None
"""
import json
import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48
CRC_MASK_DELTA = 0xA282EAD8
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"

# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
          9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_STRING, DT_BFLOAT16 = 7, 14


class BundleError(ValueError):
    """the file is not what this reader implements (or is damaged)"""


# ---------------------------------------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli), table-driven
# ---------------------------------------------------------------------------------------------------------------------------------
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TAB
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


# Large tensors: the CRC is linear over GF(2), so the buffer is cut into 2^k equal chunks whose CRCs advance together (one numpy table
# lookup per byte POSITION instead of per byte), and neighbouring chunk CRCs are then folded pairwise with the "append n zero bytes"
# operator of zlib's crc32_combine - ~100 MB/s instead of ~5.
def _gf2_times(mat, vec: int) -> int:
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_square(mat):
    return [_gf2_times(mat, mat[i]) for i in range(32)]


def _zeros_operator(nbytes: int):
    """the 32 x 32 GF(2) matrix (as 32 column words) that advances a CRC register over nbytes zero bytes"""
    odd = [0x82F63B78] + [1 << i for i in range(31)]        # one zero BIT
    op = None
    cur = _gf2_square(_gf2_square(_gf2_square(odd)))         # 8 bits = one byte
    n = nbytes
    while n:
        if n & 1:
            op = cur if op is None else [_gf2_times(cur, op[i]) for i in range(32)]
        n >>= 1
        if n:
            cur = _gf2_square(cur)
    return op if op is not None else [1 << i for i in range(32)]


def crc32c_combine(crc1: int, crc2: int, len2: int) -> int:
    """crc32c(A + B) from crc32c(A), crc32c(B), len(B)"""
    return _gf2_times(_zeros_operator(len2), crc1) ^ crc2 if len2 else crc1


def crc32c_fast(data) -> int:
    """crc32c of a bytes-like / uint8 array of any size (numpy-vectorised above 64 KB)"""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    n = a.shape[0]
    if n < (1 << 16):
        return crc32c(a.tobytes())
    k = max(1, min(16, int(np.log2(n // 1024))))            # 2^k chunks of >= 1 KB
    chunks = 1 << k
    length = n // chunks
    body = a[:chunks * length].reshape(chunks, length)
    tab = np.asarray(_CRC_TAB, dtype=np.uint32)
    c = np.full(chunks, 0xFFFFFFFF, dtype=np.uint32)
    for j in range(length):
        c = tab[(c ^ body[:, j]) & 0xFF] ^ (c >> np.uint32(8))
    c ^= np.uint32(0xFFFFFFFF)
    op, seg = _zeros_operator(length), length
    while c.shape[0] > 1:                                    # fold pairs: crc(A + B) = op . crc(A) ^ crc(B)
        left, right = c[0::2], c[1::2]
        moved = np.zeros_like(left)
        for bit in range(32):
            moved ^= np.where((left >> np.uint32(bit)) & np.uint32(1), np.uint32(op[bit]), np.uint32(0))
        c = moved ^ right
        op, seg = _gf2_square(op), seg * 2
    total = int(c[0])
    tail = a[chunks * length:]
    return crc32c_combine(total, crc32c(tail.tobytes()), tail.shape[0]) if tail.shape[0] else total


def crc_mask(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + CRC_MASK_DELTA) & 0xFFFFFFFF


def crc_unmask(m: int) -> int:
    r = (m - CRC_MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------------------------
# varints / protobuf wire format
# ---------------------------------------------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise BundleError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise BundleError("varint longer than 64 bits")


def proto_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one protobuf message: varints as int, fixed32 / fixed64 as int, length-delimited as bytes"""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > len(buf):
                raise BundleError("truncated fixed64")
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise BundleError("truncated length-delimited field")
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            if pos + 4 > len(buf):
                raise BundleError("truncated fixed32")
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise BundleError(f"protobuf wire type {wt} (groups) is not supported")
        yield field, wt, val


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


# ---------------------------------------------------------------------------------------------------------------------------------
# the table file (LevelDB format, tensorflow/core/lib/io/{format,block,table}.cc)
# ---------------------------------------------------------------------------------------------------------------------------------
def _block(buf: bytes, offset: int, size: int, what: str) -> bytes:
    """contents of the block at (offset, size); the 5-byte trailer (type, masked crc32c of contents + type) is checked"""
    if offset < 0 or size < 0 or offset + size + 5 > len(buf):
        raise BundleError(f"{what}: block handle ({offset}, {size}) outside the file")
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
    if crc_unmask(stored) != crc32c(bytes([ctype]), crc32c(contents)):
        raise BundleError(f"{what}: block checksum mismatch")
    if ctype != 0:
        raise BundleError(f"{what}: compressed block (type {ctype}); TensorFlow writes bundle indices uncompressed, snappy is not implemented")
    return contents


def _block_entries(block: bytes, what: str) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise BundleError(f"{what}: block shorter than its restart count")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    if n_restarts == 0 or end < 0:
        raise BundleError(f"{what}: bad restart array")
    out, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise BundleError(f"{what}: entry runs past the block")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path: str) -> List[Tuple[bytes, bytes]]:
    """all (key, value) pairs of a table file, in key order"""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < FOOTER_LEN:
        raise BundleError(f"{path}: shorter than a table footer")
    footer = buf[-FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise BundleError(f"{path}: not a TensorFlow table file (bad magic number)")
    pos = 0
    _mi_off, pos = _varint(footer, pos)
    _mi_size, pos = _varint(footer, pos)
    ix_off, pos = _varint(footer, pos)
    ix_size, pos = _varint(footer, pos)
    out: List[Tuple[bytes, bytes]] = []
    last = None
    for _sep, handle in _block_entries(_block(buf, ix_off, ix_size, f"{path} index"), f"{path} index"):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for k, v in _block_entries(_block(buf, off, size, f"{path} data"), f"{path} data"):
            if last is not None and k <= last:
                raise BundleError(f"{path}: keys out of order")
            last = k
            out.append((k, v))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# the bundle (tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/util/tensor_bundle/tensor_bundle.cc)
# ---------------------------------------------------------------------------------------------------------------------------------
def _shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for f, wt, v in proto_fields(buf):
        if f == 2 and wt == 2:                       # Dim
            size = 0
            for f2, wt2, v2 in proto_fields(v):
                if f2 == 1 and wt2 == 0:
                    size = _signed64(v2)
            if size < 0:
                raise BundleError("tensor with an unknown dimension")
            dims.append(size)
        elif f == 3 and wt == 0 and v:
            raise BundleError("tensor of unknown rank")
    return tuple(dims)


class _Entry:
    __slots__ = ("dtype", "shape", "shard", "offset", "size", "crc", "sliced")

    def __init__(self, buf: bytes):
        self.dtype, self.shape, self.shard, self.offset, self.size, self.crc, self.sliced = 0, (), 0, 0, 0, None, False
        for f, wt, v in proto_fields(buf):
            if f == 1 and wt == 0:
                self.dtype = v
            elif f == 2 and wt == 2:
                self.shape = _shape(v)
            elif f == 3 and wt == 0:
                self.shard = v
            elif f == 4 and wt == 0:
                self.offset = v
            elif f == 5 and wt == 0:
                self.size = v
            elif f == 6 and wt == 5:
                self.crc = v
            elif f == 7:
                self.sliced = True


class Bundle:
    """a checkpoint ``<prefix>.index`` + ``<prefix>.data-XXXXX-of-XXXXX``"""

    def __init__(self, prefix: str):
        self.prefix = prefix
        pairs = read_table(prefix + ".index")
        if not pairs or pairs[0][0] != b"":
            raise BundleError(f"{prefix}.index: no bundle header (the empty key)")
        self.num_shards, endian = 1, 0
        for f, wt, v in proto_fields(pairs[0][1]):
            if f == 1 and wt == 0:
                self.num_shards = v
            elif f == 2 and wt == 0:
                endian = v
        if endian != 0:
            raise BundleError(f"{prefix}.index: big-endian bundle")
        self.entries: Dict[str, _Entry] = {k.decode("utf-8"): _Entry(v) for k, v in pairs[1:]}
        self._shards: Dict[int, np.memmap] = {}

    def _data(self, shard: int):
        if shard not in self._shards:
            path = f"{self.prefix}.data-{shard:05d}-of-{self.num_shards:05d}"
            if not os.path.isfile(path):
                raise BundleError(f"{path}: data shard not found")
            self._shards[shard] = np.memmap(path, dtype=np.uint8, mode="r")
        return self._shards[shard]

    def _bytes(self, e: _Entry, name: str) -> bytes:
        if e.sliced:
            raise BundleError(f"{name}: partitioned (sliced) variable")
        data = self._data(e.shard)
        if e.offset + e.size > data.shape[0]:
            raise BundleError(f"{name}: tensor bytes run past the data shard")
        raw = bytes(data[e.offset:e.offset + e.size])
        if e.crc is not None and e.dtype != DT_STRING:
            # tensor_bundle.cc stores the MASKED crc32c of the tensor bytes; the plain value is accepted as well (this reader has never
            # met TensorFlow's own output, and a false refusal of a good file helps nobody - a damaged tensor matches neither)
            c = crc32c_fast(raw)
            if c != crc_unmask(e.crc) and c != e.crc:
                raise BundleError(f"{name}: tensor checksum mismatch")
        return raw

    def tensor(self, name: str) -> np.ndarray:
        e = self.entries[name]
        if e.dtype == DT_STRING:
            raise BundleError(f"{name}: string tensor (use string_scalar)")
        raw = self._bytes(e, name)
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if e.dtype == DT_BFLOAT16:
            if len(raw) != 2 * n:
                raise BundleError(f"{name}: {len(raw)} bytes for {n} bfloat16 values")
            return (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32).reshape(e.shape)
        dt = DTYPES.get(e.dtype)
        if dt is None:
            raise BundleError(f"{name}: dtype enum {e.dtype} is not supported")
        if len(raw) != n * dt.itemsize:
            raise BundleError(f"{name}: {len(raw)} bytes for shape {e.shape} of {dt}")
        return np.frombuffer(raw, dtype=dt).reshape(e.shape).copy()

    def string_scalar(self, name: str) -> bytes:
        """a scalar DT_STRING tensor: varint64 length, 4-byte checksum of the lengths, the bytes (tensor_bundle.cc WriteStringTensor)"""
        e = self.entries[name]
        if e.dtype != DT_STRING or e.shape not in ((), (1,)):
            raise BundleError(f"{name}: not a scalar string tensor")
        raw = self._bytes(e, name)
        n, pos = _varint(raw, 0)
        pos += 4
        if pos + n != len(raw):
            raise BundleError(f"{name}: string length {n} does not match the entry size {len(raw)}")
        return raw[pos:pos + n]

    def object_graph_names(self) -> Dict[str, str]:
        """{checkpoint key: variable name} from the object graph the checkpoint carries (trackable_object_graph.proto: every node's
        attributes hold the variable's ``full_name`` and its ``checkpoint_key``); keys without a recorded name are left out"""
        if OBJECT_GRAPH_KEY not in self.entries:
            return {}
        out: Dict[str, str] = {}
        for f, wt, node in proto_fields(self.string_scalar(OBJECT_GRAPH_KEY)):
            if f != 1 or wt != 2:
                continue
            for f2, wt2, attr in proto_fields(node):
                if f2 != 2 or wt2 != 2:
                    continue
                full, key = "", ""
                for f3, wt3, v in proto_fields(attr):
                    if f3 == 2 and wt3 == 2:
                        full = v.decode("utf-8")
                    elif f3 == 3 and wt3 == 2:
                        key = v.decode("utf-8")
                if full and key:
                    out[key] = full
        return out


def load_tf_checkpoint(prefix: str) -> Dict[str, np.ndarray]:
    """every variable of an object-based TF2 checkpoint under its graph name (``dense/kernel``), optimizer slots and the bookkeeping
    entries (save counter, the object graph itself) left out; a name-based TF1 checkpoint (no object graph) comes back under its keys"""
    b = Bundle(prefix)
    names = b.object_graph_names()
    out: Dict[str, np.ndarray] = {}
    unnamed = []
    for key, e in b.entries.items():
        if key == OBJECT_GRAPH_KEY or e.dtype == DT_STRING:
            continue
        if names:
            # bookkeeping that is not a model variable: optimizer state and slots, the save counter, and what Keras tracks under
            # keras_api/ (metric totals / counts) - their un-prefixed names would also defeat the common-scope stripping of the caller
            if ("/.OPTIMIZER_SLOT/" in key or key.startswith(("optimizer/", "save_counter/", "keras_api/")) or "/keras_api/" in key):
                continue
            if key not in names:
                if key.endswith("/.ATTRIBUTES/VARIABLE_VALUE"):
                    unnamed.append(key)              # a variable whose object-graph attribute carries no full_name
                continue
            name = names[key]
        else:
            name = key
        name = name[:-2] if name.endswith(":0") else name
        if name in out:
            raise BundleError(f"{prefix}: two checkpoint entries map to the variable {name!r}")
        out[name] = b.tensor(key)
    if unnamed:
        # dropping them silently would surface later as "missing variable X" from a constructor, far from the cause
        raise BundleError(f"{prefix}: {len(unnamed)} variable(s) have no name recorded in the object graph (full_name empty) and cannot be "
                          f"mapped onto the member's Keras variables: {', '.join(sorted(unnamed)[:8])}{' ...' if len(unnamed) > 8 else ''}")
    if not out:
        raise BundleError(f"{prefix}: no variables found")
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Keras SavedModel directory
# ---------------------------------------------------------------------------------------------------------------------------------
def _model_dir(path: str) -> str:
    d = os.path.dirname(path) if os.path.basename(path) == "saved_model.pb" else path
    if not os.path.isfile(os.path.join(d, "saved_model.pb")):
        raise BundleError(f"{path}: no saved_model.pb")
    return d


def load_savedmodel_weights(path: str) -> Dict[str, np.ndarray]:
    d = _model_dir(path)
    return load_tf_checkpoint(os.path.join(d, "variables", "variables"))


def load_savedmodel_config(path: str) -> Optional[dict]:
    """The Keras model config from ``keras_metadata.pb`` (saved_metadata.proto: repeated SavedObject nodes = 1 with node_path = 3,
    identifier = 4, metadata = 5 - a JSON string): the root node's ``class_name`` / ``config``, in the form ``model_config`` has in a
    Keras ``.h5`` file; None if the directory has no metadata file or no model node."""
    d = _model_dir(path)
    meta = os.path.join(d, "keras_metadata.pb")
    if not os.path.isfile(meta):
        return None
    with open(meta, "rb") as f:
        buf = f.read()
    for f1, wt, node in proto_fields(buf):
        if f1 != 1 or wt != 2:
            continue
        node_path, ident, md = "", "", ""
        for f2, wt2, v in proto_fields(node):
            if wt2 != 2:
                continue
            if f2 == 3:
                node_path = v.decode("utf-8")
            elif f2 == 4:
                ident = v.decode("utf-8")
            elif f2 == 5:
                md = v.decode("utf-8")
        if node_path == "root" and ident in ("_tf_keras_model", "_tf_keras_network", "_tf_keras_sequential") and md:
            j = json.loads(md)
            if isinstance(j, dict) and "config" in j:
                return {"class_name": j.get("class_name", "Functional"), "config": j["config"]}
    return None
