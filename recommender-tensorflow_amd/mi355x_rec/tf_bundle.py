"""TensorFlow tensor-bundle checkpoints (``model.ckpt-N.index`` + ``model.ckpt-N.data-?????-of-?????``), read and written
in pure Python — the files ``tf.estimator`` writes under the reference's ``RunConfig`` (trainers/conf_utils.py:6-10).

Why: the engine imports a reference checkpoint BY VARIABLE NAME (``tf_names.py``).  Until round 3 that needed a
dump made with TensorFlow itself; with this module ``--warm-start-from <model_dir | checkpoint prefix>`` reads the bundle
directly, and ``write_bundle`` produces one (for tests, and for handing a model trained here back to TF code).

Format, restated from TensorFlow's ``tensor_bundle`` / ``lib/io/table`` sources **[TF-1.12, recalled — no TF-written bundle
exists in this container to check against: the reader is tested on bundles the writer half produced; parity unpinned]**:

* ``.index`` is a leveldb-style sorted table: data blocks of prefix-compressed (key, value) entries with restart points
  every 16 entries, each block followed by a 5-byte trailer (compression type 0 + masked CRC-32C of block and type), a
  metaindex block, an index block (separator key -> block handle) and a 48-byte footer ending in the magic
  0xdb4775248b80fb57.  Key "" holds a ``BundleHeaderProto`` (num_shards, endianness, version); every other key is a
  tensor name holding a ``BundleEntryProto`` (dtype, shape, shard_id, offset, size, masked crc32c of the bytes).
* ``.data-SSSSS-of-NNNNN`` holds the tensors' raw little-endian bytes at those offsets.

CRC-32C comes from the library's host entry ``mi_crc32c`` (slicing-by-8: a 6.6 GB embedding table checks in seconds).
Not supported (raises, naming the variable): snappy-compressed index blocks, big-endian bundles, partitioned variables
(entries with ``slices``), string / resource / variant tensors."""
import glob
import os
import re
import struct

import numpy as np

from . import _lib

MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto DataType
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


def crc32c(data, crc=0):
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    if buf.size == 0:
        return crc
    buf = np.ascontiguousarray(buf)
    return int(_lib.load().mi_crc32c(buf.ctypes.data, buf.size, crc))


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


# ---------------------------------------------------------------------------------------- varints / tiny protobuf
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if b < 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of a serialized message; value = int (varint / fixed) or bytes (length-delimited)"""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("protobuf wire type %d" % wt)
        yield num, wt, v


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _entry_proto(dtype, shape, shard_id, offset, size, crc_masked):
    out = bytearray()
    out += b"\x08"; _put_varint(out, dtype)                              # 1: dtype
    sh = bytearray()
    for d in shape:                                                      # TensorShapeProto.dim = 2 { size = 1 }
        dim = bytearray(b"\x08"); _put_varint(dim, int(d))
        sh += b"\x12"; _put_varint(sh, len(dim)); sh += dim
    out += b"\x12"; _put_varint(out, len(sh)); out += sh                 # 2: shape
    if shard_id:
        out += b"\x18"; _put_varint(out, shard_id)                       # 3: shard_id
    if offset:
        out += b"\x20"; _put_varint(out, offset)                         # 4: offset
    out += b"\x28"; _put_varint(out, size)                               # 5: size
    out += b"\x35" + struct.pack("<I", crc_masked)                       # 6: crc32c (fixed32)
    return bytes(out)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for num, wt, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, v2 in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in _fields(v2):
                        if n3 == 1:
                            size = _signed(v3)
                    e["shape"].append(size)
                elif n2 == 3 and v2:
                    raise NotImplementedError("tensor of unknown rank in a bundle")
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


def _header_proto(num_shards):
    out = bytearray(b"\x08"); _put_varint(out, num_shards)               # 1: num_shards; 2: endianness LITTLE = 0 (default)
    ver = bytearray(b"\x08"); _put_varint(ver, 1)                        # 3: version { producer = 1 } (kTensorBundleVersion)
    out += b"\x1a"; _put_varint(out, len(ver)); out += ver
    return bytes(out)


def _parse_header(buf):
    h = {"num_shards": 0, "endianness": 0}
    for num, _, v in _fields(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
    return h


# ---------------------------------------------------------------------------------------- the sorted table (.index)
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.count and self.count % self.ri == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        _put_varint(self.buf, shared); _put_varint(self.buf, len(key) - shared); _put_varint(self.buf, len(value))
        self.buf += key[shared:]; self.buf += value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_block(f, contents):
    """block + trailer (type 0 = uncompressed, masked crc32c of contents + type); returns the (offset, size) handle"""
    off = f.tell()
    f.write(contents)
    f.write(b"\x00" + struct.pack("<I", mask(crc32c(b"\x00", crc32c(contents)))))
    return off, len(contents)


def _handle(off, size):
    out = bytearray(); _put_varint(out, off); _put_varint(out, size)
    return bytes(out)


def _write_table(path, items, block_size=4096):
    """items: (key bytes, value bytes) sorted by key"""
    with open(path, "wb") as f:
        index, blk = _BlockBuilder(1), _BlockBuilder()
        for key, value in items:
            blk.add(key, value)
            if blk.size() >= block_size:
                index.add(blk.last, _handle(*_write_block(f, blk.finish())))      # (the block's last key is a valid separator)
                blk = _BlockBuilder()
        if blk.count:
            index.add(blk.last, _handle(*_write_block(f, blk.finish())))
        meta = _handle(*_write_block(f, _BlockBuilder().finish()))
        idx = _handle(*_write_block(f, index.finish()))
        foot = meta + idx
        f.write(foot + b"\x00" * (40 - len(foot)) + struct.pack("<II", MAGIC & 0xffffffff, MAGIC >> 32))


def _read_block(data, off, size, what):
    contents, trailer = data[off:off + size], data[off + size:off + size + 5]
    if len(trailer) != 5:
        raise ValueError("%s: block at %d runs past the end of the file" % (what, off))
    if trailer[0] != 0:
        raise NotImplementedError("%s: compressed table block (type %d); TensorFlow writes bundle indexes uncompressed" % (what, trailer[0]))
    if mask(crc32c(trailer[:1], crc32c(contents))) != struct.unpack("<I", trailer[1:])[0]:
        raise ValueError("%s: checksum mismatch in the table block at offset %d" % (what, off))
    n_restarts = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    end = len(contents) - 4 * (n_restarts + 1)
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(contents, pos)
        non_shared, pos = _get_varint(contents, pos)
        vlen, pos = _get_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen]))); pos += vlen
    return out


def _read_table(path):
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<II", data[-8:]) != (MAGIC & 0xffffffff, MAGIC >> 32):
        raise ValueError("%s is not a TensorFlow table file (bad magic)" % path)
    foot = data[-48:-8]
    _, p = _get_varint(foot, 0); _, p = _get_varint(foot, p)                  # metaindex handle (unused)
    ioff, p = _get_varint(foot, p); isize, p = _get_varint(foot, p)
    items = []
    for _, hv in _read_block(data, ioff, isize, path):
        boff, q = _get_varint(hv, 0); bsize, q = _get_varint(hv, q)
        items += _read_block(data, boff, bsize, path)
    return items


# ---------------------------------------------------------------------------------------- bundles
def _prefix_of(path):
    """a checkpoint prefix from a prefix, an .index file, or a model_dir (its newest model.ckpt-N / its `checkpoint` file)"""
    if os.path.isdir(path):
        ck = os.path.join(path, "checkpoint")
        if os.path.exists(ck):
            m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(ck).read())
            if m:
                p = m.group(1)
                return p if os.path.isabs(p) else os.path.join(path, p)
        idx = glob.glob(os.path.join(path, "*.index"))
        if not idx:
            raise FileNotFoundError("no checkpoint (*.index) under %s" % path)
        step = lambda s: int(re.search(r"-(\d+)\.index$", s).group(1)) if re.search(r"-(\d+)\.index$", s) else -1
        return max(idx, key=step)[:-len(".index")]
    return path[:-len(".index")] if path.endswith(".index") else path


def is_bundle(path):
    try:
        return os.path.exists(_prefix_of(path) + ".index")
    except (FileNotFoundError, OSError):
        return False


def list_variables(path):
    """name -> (numpy dtype, shape) without reading the data shards"""
    out = {}
    for key, value in _read_table(_prefix_of(path) + ".index"):
        if key:
            e = _parse_entry(value)
            out[key.decode()] = (_DTYPES.get(e["dtype"]), tuple(e["shape"]))
    return out


def read_bundle(path, names=None, verify=True):
    """{variable name: ndarray} of a TensorFlow checkpoint (all variables, or `names`).  Every tensor's bytes are checked
    against the crc32c its index entry carries (verify=False skips that)."""
    prefix = _prefix_of(path)
    items = _read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    head = _parse_header(items[0][1])
    if head["endianness"] != 0:
        raise NotImplementedError("big-endian tensor bundle")
    want = None if names is None else set(names)
    shards, out = {}, {}
    for key, value in items[1:]:
        name = key.decode()
        if want is not None and name not in want:
            continue
        e = _parse_entry(value)
        if e["slices"]:
            raise NotImplementedError("variable %r is partitioned (stored as slices): not supported" % name)
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            raise NotImplementedError("variable %r has TensorFlow dtype %d (string / resource / ...): not supported" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, head["num_shards"]), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if raw.size != e["size"] or e["size"] != count * np.dtype(dt).itemsize:
            raise ValueError("variable %r: %d bytes in the data shard, shape %s of %s needs %d" %
                             (name, raw.size, e["shape"], np.dtype(dt).name, count * np.dtype(dt).itemsize))
        arr = np.array(raw)                                               # out of the memory map
        if verify and e["crc32c"] is not None and mask(crc32c(arr)) != e["crc32c"]:
            raise ValueError("variable %r: checksum mismatch in %s" % (name, shards[sid].filename))
        out[name] = arr.view(dt).reshape(e["shape"])
    if want is not None and want - set(out):
        raise KeyError("not in the checkpoint: %s" % sorted(want - set(out))[:4])
    return out


def write_bundle(prefix, tensors, block_size=4096):
    """Writes {name: ndarray} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little endian) and a
    `checkpoint` state file beside them, the way tf.train.Saver lays a V2 checkpoint out."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items, off = [(b"", _header_proto(1))], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:                     # (np.ascontiguousarray would turn a scalar into shape (1,))
                a = np.ascontiguousarray(a)
            if a.dtype not in _DT_OF:
                raise NotImplementedError("dtype %s of %r" % (a.dtype, name))
            raw = a.reshape(-1).view(np.uint8)
            f.write(raw.tobytes())
            items.append((name.encode(), _entry_proto(_DT_OF[a.dtype], a.shape, 0, off, raw.size, mask(crc32c(raw)))))
            off += raw.size
    _write_table(prefix + ".index", items, block_size)
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix
