"""Read-only HDF5 subset in pure Python + numpy - enough for Keras ``.h5`` weight / model files.

The reference loads its checkpoints with ``tf.keras.models.load_model(path)`` on ``*.h5`` files (main.py:101-107,186-194); this image
has neither h5py nor TensorFlow for the interpreter the product runs on, so the file format is read directly (HDF5 File Format
Specification 2.0 / 3.0, the parts libhdf5 1.8-1.12 writes with its default "earliest" format bounds, which is what h5py / Keras use):

* superblock versions 0-3; "old style" groups (symbol-table message -> v1 B-tree of SNOD nodes + local heap) and compact "new style"
  groups (link messages); object headers version 1 and 2 with continuation blocks;
* datasets: compact, contiguous and chunked (v1 chunk B-tree) layouts; filters deflate, shuffle, fletcher32; fixed-point, IEEE float
  (f2/f4/f8) and fixed-length string types, little or big endian;
* attributes (message versions 1-3) with the same types plus variable-length strings (global heap).
Anything else (dense groups / fractal heaps, v2 chunk indices, compound types, external files, ...) raises ``H5Unsupported`` with
the name of the structure - never a silent wrong answer.

Pinned by ``tests/test_h5lite_cpu.py`` on files written by h5py 3.3 / libhdf5 1.10.6 in the Keras layout (``tools/make_h5_fixtures.py``).
"""
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class H5Unsupported(H5Error):
    pass


class _Reader:
    def __init__(self, buf: bytes, so: int = 8, sl: int = 8):
        self.b, self.so, self.sl = buf, so, sl

    def u(self, off: int, n: int) -> int:
        if off < 0 or off + n > len(self.b):
            raise H5Error(f"read of {n} bytes at {off} past the end of the file ({len(self.b)} bytes)")
        return int.from_bytes(self.b[off:off + n], "little")

    def offs(self, off: int) -> int:
        v = self.u(off, self.so)
        return UNDEF if v == (1 << (8 * self.so)) - 1 else v

    def lens(self, off: int) -> int:
        return self.u(off, self.sl)


def _pad8(n: int) -> int:
    return (n + 7) & ~7


class Dataset:
    def __init__(self, f: "File", name: str, msgs):
        self.file, self.name, self._msgs = f, name, msgs
        self.shape: Tuple[int, ...] = ()
        self.dtype: Optional[np.dtype] = None
        self.attrs: Dict[str, object] = {}
        layout = filters = None
        for t, off, size, flags in msgs:
            if t == 0x1:
                self.shape = f._dataspace(off)
            elif t == 0x3:
                self.dtype, self._vlen = f._datatype(off)
            elif t == 0x8:
                layout = (off, size)
            elif t == 0xB:
                filters = f._filters(off)
            elif t == 0xC:
                k, v = f._attribute(off)
                self.attrs[k] = v
            elif t == 0x15:
                f._refuse_dense_attributes(off, name)
        if layout is None or self.dtype is None:
            raise H5Error(f"{name}: dataset without a layout or datatype message")
        self._layout, self._filters = layout, filters or []

    def __getitem__(self, _ellipsis) -> np.ndarray:
        return self.read()

    def read(self) -> np.ndarray:
        r = self.file.r
        if self._vlen:
            raise H5Unsupported(f"{self.name}: variable-length dataset")
        off, _ = self._layout
        ver = r.u(off, 1)
        n_el = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        nbytes = n_el * self.dtype.itemsize
        if n_el < 0 or nbytes > 64 * max(len(r.b), 1 << 20):     # deflate cannot expand by more: a damaged dataspace, not data
            raise H5Error(f"{self.name}: shape {self.shape} is not plausible for a {len(r.b)}-byte file")
        if ver in (3, 4):           # version 4 (libver "latest") keeps the compact / contiguous forms; its chunk indices differ
            cls = r.u(off + 1, 1)
            if ver == 4 and cls == 2:
                raise H5Unsupported(f"{self.name}: version-4 chunked layout (chunk index types of the 1.10 format)")
            if cls == 0:        # compact
                size = r.u(off + 2, 2)
                data = r.b[off + 4:off + 4 + size]
                return np.frombuffer(data[:nbytes], dtype=self.dtype).reshape(self.shape).copy()
            if cls == 1:        # contiguous
                addr = r.offs(off + 2)
                if addr == UNDEF:     # never written: fill value (zeros)
                    return np.zeros(self.shape, dtype=self.dtype)
                return np.frombuffer(r.b[addr:addr + nbytes], dtype=self.dtype).reshape(self.shape).copy()
            if cls == 2:        # chunked, v1 B-tree
                nd = r.u(off + 2, 1)
                bt = r.offs(off + 3)
                cdims = [r.u(off + 3 + r.so + 4 * i, 4) for i in range(nd)]
                return self._read_chunked(bt, cdims[:-1])
            raise H5Unsupported(f"{self.name}: data layout class {cls}")
        if ver in (1, 2):
            nd = r.u(off + 1, 1)
            cls = r.u(off + 2, 1)
            p = off + 8
            addr = None
            if cls != 0:
                addr = r.offs(p)
                p += r.so
            dims = [r.u(p + 4 * i, 4) for i in range(nd)]
            p += 4 * nd
            if cls == 1:
                return np.frombuffer(r.b[addr:addr + nbytes], dtype=self.dtype).reshape(self.shape).copy()
            if cls == 2:
                return self._read_chunked(addr, dims[:-1])
            if cls == 0:
                size = r.u(p, 4)
                return np.frombuffer(r.b[p + 4:p + 4 + size][:nbytes], dtype=self.dtype).reshape(self.shape).copy()
        raise H5Unsupported(f"{self.name}: data layout message version {ver}")

    def _read_chunked(self, btree: int, cdims: List[int]) -> np.ndarray:
        out = np.zeros(self.shape, dtype=self.dtype)
        if btree == UNDEF:
            return out
        rank = len(self.shape)
        for size, mask, coords, addr in self.file._chunk_leaves(btree, rank):
            raw = self.file.r.b[addr:addr + size]
            for i, (fid, cdata) in reversed(list(enumerate(self._filters))):
                if mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cdata[0] if cdata else self.dtype.itemsize
                    n = len(raw) // es
                    raw = np.frombuffer(raw[:n * es], dtype=np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
                elif fid == 3:
                    raw = raw[:-4]
                else:
                    raise H5Unsupported(f"{self.name}: filter id {fid}")
            chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl_out = tuple(slice(c, min(c + d, s)) for c, d, s in zip(coords, cdims, self.shape))
            sl_in = tuple(slice(0, so.stop - so.start) for so in sl_out)
            out[sl_out] = chunk[sl_in]
        return out


class Group:
    def __init__(self, f: "File", name: str, msgs):
        self.file, self.name = f, name
        self.attrs: Dict[str, object] = {}
        self._links: Dict[str, int] = {}
        for t, off, size, flags in msgs:
            if t == 0x11:      # symbol table: old-style group
                for k, addr in f._symbol_table(f.r.offs(off), f.r.offs(off + f.r.so)):
                    self._links[k] = addr
            elif t == 0x6:     # link message: compact new-style group
                k, addr = f._link(off)
                if addr is not None:
                    self._links[k] = addr
            elif t == 0x2:     # link info: dense storage if the fractal heap address is defined
                fl = f.r.u(off + 1, 1)
                p = off + 2 + (8 if fl & 1 else 0)
                if f.r.offs(p) != UNDEF:
                    raise H5Unsupported(f"{name}: dense link storage (fractal heap)")
            elif t == 0xC:
                k, v = f._attribute(off)
                self.attrs[k] = v
            elif t == 0x15:
                f._refuse_dense_attributes(off, name)

    def keys(self) -> List[str]:
        return list(self._links)

    def __contains__(self, k: str) -> bool:
        return k in self._links

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(f"{path!r}: no {part!r} in {node.name!r}")
            node = node.file._object(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def visit_datasets(self, prefix: str = "") -> Iterator[Tuple[str, Dataset]]:
        """(path relative to this group, dataset) for every dataset below it, depth first in link order"""
        for k in self.keys():
            node = self[k]
            if isinstance(node, Dataset):
                yield prefix + k, node
            else:
                yield from node.visit_datasets(prefix + k + "/")


class File(Group):
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            buf = fh.read()
        base = 0
        while buf[base:base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                raise H5Error(f"{path}: not an HDF5 file")
        r0 = _Reader(buf)
        ver = r0.u(base + 8, 1)
        if ver in (0, 1):
            so, sl = r0.u(base + 13, 1), r0.u(base + 14, 1)
            self.r = _Reader(buf, so, sl)
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self.r.offs(p)
            root_ste = p + 4 * so
            root_addr = self.r.offs(root_ste + so)
        elif ver in (2, 3):
            so, sl = r0.u(base + 9, 1), r0.u(base + 10, 1)
            self.r = _Reader(buf, so, sl)
            self.base = self.r.offs(base + 12)
            root_addr = self.r.offs(base + 12 + 3 * so)
        else:
            raise H5Unsupported(f"{path}: superblock version {ver}")
        if self.base not in (0, UNDEF):
            raise H5Unsupported(f"{path}: non-zero base address")
        self._cache: Dict[int, object] = {}
        Group.__init__(self, self, "/", self._messages(root_addr))

    # ---- object headers ---------------------------------------------------------------------------------------------
    def _refuse_dense_attributes(self, off: int, name: str):
        """Attribute Info message (0x15): version, flags, [max creation index], fractal-heap address, name B-tree address.  A defined heap
        address means the object's attributes live in dense storage (more than 8 attributes, or written with libver='latest' phase-change
        settings) - not read here.  Refuse: silently returning an object WITHOUT its layer_names / weight_names would make the Keras
        loader fall back to 'every dataset in the tree' under the wrong keys."""
        r = self.r
        fl = r.u(off + 1, 1)
        p = off + 2 + (2 if fl & 1 else 0)
        if r.offs(p) != UNDEF:
            raise H5Unsupported(f"{name}: dense attribute storage (fractal heap)")

    def _messages(self, addr: int):
        """[(type, data offset, data size, flags)] of the object header at ``addr`` (continuation blocks followed)"""
        r = self.r
        out = []
        if r.b[addr:addr + 4] == b"OHDR":
            ver, flags = r.u(addr + 4, 1), r.u(addr + 5, 1)
            if ver != 2:
                raise H5Unsupported(f"object header version {ver}")
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szb = 1 << (flags & 3)
            size0 = r.u(p, szb)
            p += szb
            blocks = [(p, size0)]
            seen = {p}
            track = bool(flags & 0x04)
            while blocks:
                p, n = blocks.pop(0)
                end = p + n
                while p + 4 <= end:
                    t, size, fl = r.u(p, 1), r.u(p + 1, 2), r.u(p + 3, 1)
                    p += 4 + (2 if track else 0)
                    if t == 0x10:
                        coff, clen = r.offs(p), r.lens(p + r.so)
                        if r.b[coff:coff + 4] != b"OCHK":
                            raise H5Error("object header continuation without OCHK signature")
                        if coff + 4 in seen or len(seen) > 4096:       # a continuation that points back at a block already read
                            raise H5Error("object header continuation chain loops")
                        seen.add(coff + 4)
                        blocks.append((coff + 4, clen - 8))
                    elif t != 0:
                        out.append((t, p, size, fl))
                    p += size
            return out
        ver = r.u(addr, 1)
        if ver != 1:
            raise H5Error(f"object header at {addr}: version byte {ver}")
        nmsg, size0 = r.u(addr + 2, 2), r.u(addr + 8, 4)
        blocks = [(addr + 16, size0)]
        while blocks and nmsg > 0:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and nmsg > 0:
                t, size, fl = r.u(p, 2), r.u(p + 2, 2), r.u(p + 4, 1)
                p += 8
                nmsg -= 1
                if t == 0x10:
                    blocks.append((r.offs(p), r.lens(p + r.so)))
                elif t != 0:
                    out.append((t, p, size, fl))
                p += size
        return out

    def _object(self, addr: int, name: str):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, *_ in msgs}
        if 0x8 in types or (0x3 in types and 0x1 in types):
            for t, off, size, fl in msgs:
                if fl & 0x2:
                    raise H5Unsupported(f"{name}: shared header message")
            node = Dataset(self, name, msgs)
        else:
            node = Group(self, name, msgs)
        self._cache[addr] = node
        return node

    # ---- groups --------------------------------------------------------------------------------------------------
    def _symbol_table(self, btree: int, heap: int):
        r = self.r
        if r.b[heap:heap + 4] != b"HEAP":
            raise H5Error("local heap signature missing")
        data = r.offs(heap + 8 + 2 * r.sl)

        def name_at(o):
            e = r.b.index(b"\0", data + o)
            return r.b[data + o:e].decode("utf-8")

        def walk(node):
            if r.b[node:node + 4] == b"SNOD":
                n = r.u(node + 6, 2)
                p = node + 8
                for _ in range(n):
                    yield name_at(r.offs(p)), r.offs(p + r.so)
                    p += 2 * r.so + 24
                return
            if r.b[node:node + 4] != b"TREE" or r.u(node + 4, 1) != 0:
                raise H5Error("group B-tree node signature / type")
            n = r.u(node + 6, 2)
            p = node + 8 + 2 * r.so + r.sl            # past the first key
            for _ in range(n):
                yield from walk(r.offs(p))
                p += r.so + r.sl

        if btree != UNDEF:
            yield from walk(btree)

    def _link(self, off: int):
        r = self.r
        fl = r.u(off + 1, 1)
        p = off + 2
        ltype = 0
        if fl & 0x08:
            ltype = r.u(p, 1)
            p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        nb = 1 << (fl & 3)
        n = r.u(p, nb)
        p += nb
        name = r.b[p:p + n].decode("utf-8")
        p += n
        return name, (r.offs(p) if ltype == 0 else None)     # soft / external links are skipped

    def _chunk_leaves(self, node: int, rank: int):
        r = self.r
        if r.b[node:node + 4] != b"TREE" or r.u(node + 4, 1) != 1:
            raise H5Error("chunk B-tree node signature / type")
        level, n = r.u(node + 5, 1), r.u(node + 6, 2)
        ksz = 8 + 8 * (rank + 1)
        p = node + 8 + 2 * r.so
        for _ in range(n):
            size, mask = r.u(p, 4), r.u(p + 4, 4)
            coords = [r.u(p + 8 + 8 * i, 8) for i in range(rank)]
            child = r.offs(p + ksz)
            if level == 0:
                yield size, mask, coords, child
            else:
                yield from self._chunk_leaves(child, rank)
            p += ksz + r.so

    # ---- messages -------------------------------------------------------------------------------------------------
    def _dataspace(self, off: int) -> Tuple[int, ...]:
        r = self.r
        ver, rank = r.u(off, 1), r.u(off + 1, 1)
        if ver == 1:
            p = off + 8
        elif ver == 2:
            if r.u(off + 3, 1) == 2:          # null dataspace
                return (0,)
            p = off + 4
        else:
            raise H5Unsupported(f"dataspace message version {ver}")
        return tuple(r.lens(p + i * r.sl) for i in range(rank))

    def _datatype(self, off: int):
        """-> (numpy dtype, vlen kind or None); vlen kind 'str' for variable-length strings"""
        r = self.r
        cv = r.u(off, 1)
        cls, bits, size = cv & 0xF, r.u(off + 1, 3), r.u(off + 4, 4)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits & 0x8 else 'u'}{size}"), None
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Unsupported(f"{size}-byte floating point type")
            return np.dtype(f"{order}f{size}"), None
        if cls == 3:
            return np.dtype(f"S{size}"), None
        if cls == 9:
            if (bits & 0xF) == 1:
                return np.dtype("V16") if r.so == 8 else np.dtype(f"V{8 + r.so}"), "str"
            raise H5Unsupported("variable-length sequence type")
        raise H5Unsupported(f"datatype class {cls}")

    def _filters(self, off: int):
        r = self.r
        ver, n = r.u(off, 1), r.u(off + 1, 1)
        p = off + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = r.u(p, 2)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = r.u(p, 2)
                p += 2
            p += 2                                  # flags
            ncd = r.u(p, 2)
            p += 2
            p += _pad8(nlen) if ver == 1 else nlen
            cdata = [r.u(p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cdata))
        return out

    def _attribute(self, off: int):
        r = self.r
        ver = r.u(off, 1)
        nsz, tsz, ssz = r.u(off + 2, 2), r.u(off + 4, 2), r.u(off + 6, 2)
        if ver == 1:
            p = off + 8
            name_p, type_p = p, p + _pad8(nsz)
            space_p = type_p + _pad8(tsz)
            data_p = space_p + _pad8(ssz)
        elif ver in (2, 3):
            if r.u(off + 1, 1) & 0x3:
                raise H5Unsupported("attribute with a shared datatype / dataspace")
            p = off + 8 + (1 if ver == 3 else 0)
            name_p, type_p = p, p + nsz
            space_p = type_p + tsz
            data_p = space_p + ssz
        else:
            raise H5Unsupported(f"attribute message version {ver}")
        name = r.b[name_p:name_p + nsz].split(b"\0", 1)[0].decode("utf-8")
        dt, vlen = self._datatype(type_p)
        shape = self._dataspace(space_p) if ssz else ()
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        raw = np.frombuffer(r.b[data_p:data_p + n * dt.itemsize], dtype=dt)
        if vlen == "str":
            vals = [self._global_heap_string(bytes(v)) for v in raw]
            return name, (vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape))
        val = raw.reshape(shape).copy()
        return name, (val[()] if shape == () else val)

    def _global_heap_string(self, ref: bytes) -> bytes:
        r = self.r
        length = int.from_bytes(ref[:4], "little")
        coll = int.from_bytes(ref[4:4 + r.so], "little")
        idx = int.from_bytes(ref[4 + r.so:8 + r.so], "little")
        if length == 0:
            return b""
        if r.b[coll:coll + 4] != b"GCOL":
            raise H5Error("global heap collection signature missing")
        end = coll + r.lens(coll + 8)
        p = coll + 8 + r.sl
        while p + 8 + r.sl <= end:
            i, size = r.u(p, 2), r.lens(p + 8)
            if i == 0:
                break
            if i == idx:
                return r.b[p + 8 + r.sl:p + 8 + r.sl + length]
            p += 8 + r.sl + _pad8(size)
        raise H5Error(f"global heap object {idx} not found")


# ---- Keras layout ------------------------------------------------------------------------------------------------
def _names(v) -> List[str]:
    arr = np.atleast_1d(v)
    return [x.decode("utf-8") if isinstance(x, (bytes, np.bytes_)) else str(x) for x in arr.tolist()]


def load_keras_weights(path: str) -> Dict[str, np.ndarray]:
    """``_load_keras_weights`` with every failure mode of a damaged file (truncation, pointers into nowhere or in circles, bad deflate
    streams, impossible shapes) turned into ``H5Error``: a corrupt checkpoint is reported, never half-read."""
    try:
        return _load_keras_weights(path)
    except H5Error:
        raise
    except (IndexError, KeyError, ValueError, OverflowError, MemoryError, RecursionError, UnicodeDecodeError, zlib.error, struct.error) as e:
        raise H5Error(f"{path}: damaged HDF5 file ({type(e).__name__}: {e})") from e


def _load_keras_weights(path: str) -> Dict[str, np.ndarray]:
    """``model.save_weights('x.h5')`` / ``model.save('x.h5')`` -> ``{variable name without ':0': array}``, in Keras' own order
    (root attribute ``layer_names``, per-layer attribute ``weight_names``, both possibly split into ``...0, ...1`` chunks -
    keras/saving/hdf5_format.py ``save_attributes_to_hdf5_group``); a full-model file keeps the same tree under ``model_weights``.
    Files without those attributes fall back to every dataset in the tree."""
    f = File(path)
    root: Group = f["model_weights"] if "model_weights" in f else f

    def chunked_attr(g: Group, key: str) -> Optional[List[str]]:
        if key in g.attrs:
            return _names(g.attrs[key])
        out, i = [], 0
        while f"{key}{i}" in g.attrs:
            out += _names(g.attrs[f"{key}{i}"])
            i += 1
        return out or None

    out: Dict[str, np.ndarray] = {}
    layers = chunked_attr(root, "layer_names")
    if layers is None:
        for name, ds in root.visit_datasets():
            out[name.rsplit(":", 1)[0]] = ds.read()
        return out
    for layer in layers:
        g = root[layer]
        for wn in chunked_attr(g, "weight_names") or []:
            out[wn.rsplit(":", 1)[0]] = g[wn].read()
    return out


def load_keras_model_config(path: str) -> Optional[dict]:
    """The ``model_config`` root attribute of a Keras full-model ``.h5`` (``model.save('x.h5')``: keras/saving/hdf5_format.py
    ``save_model_to_hdf5`` writes ``json.dumps({'class_name': ..., 'config': ...})``) as a dictionary, or None for a weight-only file
    (``model.save_weights``).  This is what ``tf.keras.models.load_model`` (main.py:107) rebuilds the graph from."""
    import json
    try:
        f = File(path)
        raw = f.attrs.get("model_config")
        if raw is None:
            return None
        if isinstance(raw, np.ndarray):
            raw = raw.item() if raw.ndim == 0 else raw.tolist()[0]
        if isinstance(raw, (bytes, np.bytes_)):
            raw = bytes(raw).decode("utf-8")
        cfg = json.loads(raw)
        if not isinstance(cfg, dict):
            raise H5Error(f"{path}: model_config is not a JSON object")
        return cfg
    except H5Error:
        raise
    except (IndexError, KeyError, ValueError, OverflowError, MemoryError, RecursionError, UnicodeDecodeError, zlib.error, struct.error) as e:
        raise H5Error(f"{path}: damaged HDF5 file ({type(e).__name__}: {e})") from e
