"""Reader / writer for TensorFlow "tensor bundle" checkpoints (what tf.train.Saver -- phiseg_model.py:144-148, 179, 525, 535 --
writes and restores: <prefix>.index + <prefix>.data-00000-of-00001), without TensorFlow.

A user of the reference owns checkpoints in this format (`model.ckpt-<step>.*`, `model_best_{dice,loss,ged,ncc}.ckpt-<step>.*`);
`phiseg.load_weights` reads them directly and `phiseg.save_weights(..., format='tf')` writes them, so that weights move
between the two code bases under the reference's variable names.  It also backs the reference helpers
`print_tensornames_in_checkpoint_file` / `get_checkpoint_weights` (tfwrapper/utils.py:171-187, pywrap_tensorflow there).

TensorFlow is not installed here, so this module restates the published on-disk format (parity with TensorFlow's own reader is
UNPINNED -- no TF-written fixture exists in the reference repository; tests/test_tf_checkpoint.py pins the pieces that have
published known answers: CRC-32C, its mask, varints, the table footer magic):

  .index   an SSTable (tensorflow/core/lib/io/table*, the LevelDB table format):
             data blocks  | index block | metaindex block | footer (48 bytes)
           block    = entries, restart array (uint32 each), uint32 number of restarts; followed on disk by a 5-byte trailer:
                      1 byte compression type (0 = none; bundles are written uncompressed) + uint32 masked CRC-32C of block + type
           entry    = varint32 shared-key-bytes, varint32 unshared-key-bytes, varint32 value-bytes, key delta, value
           footer   = metaindex handle, index handle (each: varint64 offset, varint64 size), zero padding to 40 bytes,
                      magic 0xdb4775248b80fb57 (little-endian)
           keys     ""            -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: VersionDef {1: producer}}
                    tensor name   -> BundleEntryProto   {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}, 3: shard_id, 4: offset,
                                                         5: size, 6: fixed32 masked CRC-32C of the tensor bytes}
  .data-00000-of-00001   the tensors' little-endian bytes back to back, in key order.

The checksums run through libphx's host routine (phx_crc32c); like the rest of the package this module needs the built library."""
import ctypes
import os
import struct

import numpy as np

from phiseg_code_amd import runtime as rt

MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in DTYPES.items()}
BLOCK_SIZE = 256 * 1024          # table::Options::block_size of the bundle writer
RESTART_INTERVAL = 16


def crc32c(data, crc=0):
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    c = ctypes.c_uint(crc)
    rt.lib().crc32c(buf, len(buf), ctypes.byref(c))
    return c.value


def crc32c_array(a, crc=0):
    a = np.ascontiguousarray(a)
    c = ctypes.c_uint(crc)
    rt.lib().crc32c(a.ctypes.data, a.nbytes, ctypes.byref(c))
    return c.value


def mask(crc):
    """crc32c::Mask: rotate right by 15 bits and add a constant (a CRC of bytes that contain CRCs stays well distributed)."""
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask(m):
    rot = (m - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / protobuf wire format --------------------------------------------------------------------------------------
def put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def get_varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if b < 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(buf):
    """protobuf wire format -> [(field number, wire type, value)]; value: int (varint / fixed) or bytes (length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _entry_proto(dtype, shape, offset, size, crc_masked):
    dims = b"".join(b"\x12" + put_varint(len(d)) + d for d in (b"\x08" + put_varint(int(s)) for s in shape))
    out = b"\x08" + put_varint(dtype) + b"\x12" + put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + put_varint(offset)
    out += b"\x28" + put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return out


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for num, wt, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, dim in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, dv in _fields(dim):
                        if n3 == 1:
                            size = _signed64(dv)
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


# ---- table blocks -----------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
    body, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(body) != size or len(trailer) != 5:
        raise ValueError("truncated table block at %d" % offset)
    if trailer[0] != 0:
        raise NotImplementedError("compressed table block (type %d); tensor bundles are written uncompressed" % trailer[0])
    if verify:
        want = unmask(struct.unpack("<I", trailer[1:5])[0])
        if crc32c(bytes(body) + bytes(trailer[:1])) != want:
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    nrestart = struct.unpack_from("<I", body, size - 4)[0]
    end = size - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = get_varint(body, pos)
        unshared, pos = get_varint(body, pos)
        vlen, pos = get_varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return out


def _handle(buf, pos):
    off, pos = get_varint(buf, pos)
    size, pos = get_varint(buf, pos)
    return off, size, pos


def _index_entries(prefix, verify=True):
    buf = open(prefix + ".index", "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise ValueError("%s.index is not a TensorFlow tensor-bundle index (bad table magic)" % prefix)
    footer = buf[-48:]
    _, _, pos = _handle(footer, 0)
    ioff, isize, _ = _handle(footer, pos)
    out = []
    for _, hv in _read_block(buf, ioff, isize, verify):
        boff, bsize, _ = _handle(hv, 0)
        out.extend(_read_block(buf, boff, bsize, verify))
    return out


def list_variables(prefix):
    """-> {name: (numpy dtype, shape)}  (the reader's get_variable_to_shape_map + dtypes)"""
    out = {}
    for key, val in _index_entries(prefix, verify=False):
        if key == b"":
            continue
        e = _parse_entry(val)
        out[key.decode()] = (np.dtype(DTYPES[e["dtype"]]) if e["dtype"] in DTYPES else None, tuple(e["shape"]))
    return out


def read(prefix, names=None, verify=True):
    """-> {name: ndarray} of every (or the named) tensor(s) of the bundle <prefix>.index / .data-*; verify: check the table-block
    and per-tensor CRC-32C."""
    entries = _index_entries(prefix, verify)
    header = dict(num_shards=1, endianness=0)
    for key, val in entries:
        if key == b"":
            for num, _, v in _fields(val):
                if num == 1:
                    header["num_shards"] = v
                elif num == 2:
                    header["endianness"] = v
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian tensor bundle")
    shards, out = {}, {}
    for key, val in entries:
        if key == b"":
            continue
        name = key.decode()
        if names is not None and name not in names:
            continue
        e = _parse_entry(val)
        if e["sliced"]:
            raise NotImplementedError("%s: partitioned variable (tensor slices)" % name)
        if e["dtype"] not in DTYPES:
            raise NotImplementedError("%s: dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"]), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        dt = np.dtype(DTYPES[e["dtype"]])
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if raw.size != e["size"] or n * dt.itemsize != e["size"]:
            raise ValueError("%s: size %d does not match shape %s of %s" % (name, e["size"], e["shape"], dt))
        if verify and e["crc32c"] is not None and crc32c_array(raw) != unmask(e["crc32c"]):
            raise ValueError("%s: tensor checksum mismatch" % name)
        out[name] = raw.view(dt).reshape(e["shape"]).copy()
    return out


class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _emit_block(f, body):
    off = f.tell()
    f.write(body)
    f.write(b"\x00" + struct.pack("<I", mask(crc32c(body + b"\x00"))))
    return put_varint(off) + put_varint(len(body))


def write(prefix, tensors):
    """Write {name: ndarray} as the bundle <prefix>.index + <prefix>.data-00000-of-00001 (one shard, little-endian,
    uncompressed table, keys in byte order -- the layout tf.train.Saver / BundleWriter produces)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    items = sorted(((k.encode(), np.asarray(v, order='C')) for k, v in tensors.items()), key=lambda kv: kv[0])   # (ascontiguousarray would turn scalars into 1-d)
    records = [(b"", b"\x08\x01" + b"\x1a\x02\x08\x01")]         # num_shards = 1, (little-endian = default), version.producer = 1
    with open(prefix + ".data-00000-of-00001.tmp", "wb") as f:
        off = 0
        for key, a in items:
            if not key:
                raise ValueError("empty tensor name")
            if a.dtype not in _DT_OF:
                raise NotImplementedError("%s: dtype %s" % (key.decode(), a.dtype))
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            f.write(a.tobytes())
            records.append((key, _entry_proto(_DT_OF[a.dtype], a.shape, off, a.nbytes, mask(crc32c_array(a)))))
            off += a.nbytes
    with open(prefix + ".index.tmp", "wb") as f:
        index, blk = _BlockBuilder(), _BlockBuilder()

        def flush():
            nonlocal blk
            if blk.count:
                index.add(blk.last, _emit_block(f, blk.finish()))      # (any key >= the block's last key separates it)
                blk = _BlockBuilder()
        for key, val in records:
            blk.add(key, val)
            if blk.size() >= BLOCK_SIZE:
                flush()
        flush()
        meta_h = _emit_block(f, _BlockBuilder().finish())
        index_h = _emit_block(f, index.finish())
        footer = meta_h + index_h
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    os.replace(prefix + ".data-00000-of-00001.tmp", prefix + ".data-00000-of-00001")
    os.replace(prefix + ".index.tmp", prefix + ".index")


def update_checkpoint_state(folder, basename, keep=None, all_paths=None):
    """The text file `checkpoint` tf.train.Saver maintains next to its bundles (CheckpointState proto in text format):
    model_checkpoint_path = the newest prefix, all_model_checkpoint_paths = the ones kept."""
    keep = list(all_paths or keep or [basename])
    with open(os.path.join(folder, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % basename)
        for k in keep:
            f.write('all_model_checkpoint_paths: "%s"\n' % k)
