"""TensorFlow V2 checkpoint ("tensor bundle") reader without TensorFlow.

The reference saves/restores with tf.train.Saver (train.py:95-99,166; test.py:39-42):
``<prefix>.index`` is a leveldb-format table (uncompressed blocks here) mapping variable
names to BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}; the tensors are
raw little-endian bytes in ``<prefix>.data-00000-of-00001``.  Only the ``.index`` files
ship with the reference (.MISSING_LARGE_BLOBS); `load_weights` works the moment a data
file is supplied and verifies every tensor against the index's masked CRC32C.

Host-side, pure Python/numpy.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32 = 1, 3
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), 9: np.dtype("<i8"), 2: np.dtype("<f8")}


# ---------------------------------------------------------------- crc32c (Castagnoli)
def _make_table():
    poly = 0x82F63B78
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = _make_table()
_NP_TABLE = np.array(_TABLE, dtype=np.uint32)


def crc32c(data, crc=0):
    """CRC-32C of bytes.  Table driven, pure Python: ~1 us/byte, meant for the index
    blocks and per-tensor verification on load."""
    c = crc ^ 0xFFFFFFFF
    tbl = _TABLE
    for b in bytes(data):
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    """leveldb/TF mask: rotate right by 15 and add a constant."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf
def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _proto_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message (value is an int
    for varint/fixed, bytes for length-delimited)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _parse_shape(buf):
    dims = []
    for fn, _, v in _proto_fields(buf):
        if fn == 2:                       # TensorShapeProto.dim
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


class BundleEntry:
    __slots__ = ("name", "dtype", "shape", "shard_id", "offset", "size", "crc32c")

    def __init__(self, name, buf):
        self.name, self.dtype, self.shape = name, 0, ()
        self.shard_id = self.offset = self.size = self.crc32c = 0
        for fn, _, v in _proto_fields(buf):
            if fn == 1:
                self.dtype = v
            elif fn == 2:
                self.shape = _parse_shape(v)
            elif fn == 3:
                self.shard_id = v
            elif fn == 4:
                self.offset = v
            elif fn == 5:
                self.size = v
            elif fn == 6:
                self.crc32c = v

    def __repr__(self):
        return f"<BundleEntry {self.name} dtype={self.dtype} shape={self.shape} off={self.offset} size={self.size}>"


# ---------------------------------------------------------------- leveldb table
def _read_block(data, offset, size, verify=True):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed index blocks are not supported (snappy)")
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if masked_crc(data[offset:offset + size + 1]) != stored:
            raise ValueError(f"index block at {offset}: CRC mismatch")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path, verify=True):
    """Parse ``<prefix>.index``.  Returns (header dict, {name: BundleEntry})."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a leveldb table (bad magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)   # metaindex handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries, header = {}, {}
    for _, handle in _read_block(data, ioff, isize, verify):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _read_block(data, boff, bsize, verify):
            if key == b"":
                for fn, _, v in _proto_fields(val):     # BundleHeaderProto
                    header[{1: "num_shards", 2: "endianness", 3: "version"}.get(fn, fn)] = v
            else:
                name = key.decode()
                entries[name] = BundleEntry(name, val)
    return header, entries


def model_variables(entries, model="pwcdcnet"):
    """The conv kernels/biases of the model, ignoring optimizer slots and counters
    (select by name)."""
    out = {}
    for name, e in entries.items():
        if not name.startswith(model + "/"):
            continue
        if name.endswith("/kernel") or name.endswith("/bias"):
            out[name] = e
    return out


def load_weights(prefix, model="pwcdcnet", verify=True):
    """{variable name: float32 array} from ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``."""
    _, entries = read_index(prefix + ".index", verify)
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.exists(data_path):
        raise FileNotFoundError(
            f"{data_path} is missing (the reference repository ships only the .index files); "
            "supply the checkpoint's data file next to the index")
    out = {}
    with open(data_path, "rb") as f:
        for name, e in sorted(model_variables(entries, model).items()):
            f.seek(e.offset)
            raw = f.read(e.size)
            if len(raw) != e.size:
                raise ValueError(f"{name}: data file truncated")
            if verify and masked_crc(raw) != e.crc32c:
                raise ValueError(f"{name}: CRC32C mismatch against the index")
            out[name] = np.frombuffer(raw, dtype=_DTYPES[e.dtype]).reshape(e.shape).astype(np.float32)
    return out


# ---------------------------------------------------------------- writer
def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _block(entries):
    """leveldb block without prefix compression, restart point every 16 entries."""
    body, restarts = bytearray(), []
    for i, (k, v) in enumerate(entries):
        if i % 16 == 0:
            restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(v)) + k + v
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    msg = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        msg += b"\x20" + _put_varint(offset)
    msg += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return msg


def save_weights(prefix, weights):
    """Write {name: array} as a TF V2 tensor bundle (one shard) that `load_weights` --
    and tf.train.Saver.restore -- can read back."""
    names = sorted(weights)
    entries, offset = [(b"", b"\x08\x01\x1a\x02\x08\x01")], 0   # header: 1 shard, little endian, version 1
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.ascontiguousarray(weights[n], dtype="<f4")
            raw = a.tobytes()
            f.write(raw)
            entries.append((n.encode(), _entry_proto(DT_FLOAT, a.shape, offset, len(raw), masked_crc(raw))))
            offset += len(raw)
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00")
        out.extend(struct.pack("<I", masked_crc(block + b"\x00")))
        return _put_varint(off) + _put_varint(len(block))

    data_handle = emit(_block(entries))
    meta_handle = emit(_block([]))
    index_handle = emit(_block([(entries[-1][0] + b"\x00", data_handle)]))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
