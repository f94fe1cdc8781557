"""TensorFlow tensor-bundle checkpoints (phiseg_code_amd/tfwrapper/tf_checkpoint.py) -- what the reference's tf.train.Saver
(phiseg_model.py:144-148, 179, 525, 535) writes and restores.  TensorFlow is absent: the format pieces with published known
answers are pinned here (CRC-32C check value, LevelDB's crc mask example, protobuf varints, the table magic), the rest by
round trips through the reader, including a hand-assembled index written byte by byte from the format description (an
independent second encoding, not produced by the module's writer).  CPU-only except for the model round trip."""
import os
import struct

import numpy as np
import pytest

from phiseg_code_amd.tfwrapper import tf_checkpoint as tfc


def test_crc32c_known_answers():
    assert tfc.crc32c(b"123456789") == 0xE3069283                  # the CRC-32C check value (RFC 3720 appendix B.4 family)
    assert tfc.crc32c(b"\x00" * 32) == 0x8A9136AA                    # RFC 3720 B.4: 32 bytes of zeros
    assert tfc.crc32c(b"\xff" * 32) == 0x62A8AB43                    # ... 32 bytes of ones
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E                # ... incrementing
    assert tfc.crc32c(b"6789", tfc.crc32c(b"12345")) == 0xE3069283   # running value
    a = np.arange(1000, dtype=np.float32)
    assert tfc.crc32c_array(a) == tfc.crc32c(a.tobytes()) == tfc.crc32c_array(a[1:], tfc.crc32c(a[:1].tobytes()))
    for c in (0, 1, 0xE3069283, 0xffffffff):
        assert tfc.unmask(tfc.mask(c)) == c and tfc.mask(c) != c
    assert tfc.mask(0) == 0xa282ead8


def test_varints_and_entry_proto():
    for v, enc in ((0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"), (2 ** 32, b"\x80\x80\x80\x80\x10")):
        assert tfc.put_varint(v) == enc and tfc.get_varint(enc, 0) == (v, len(enc))
    e = tfc._entry_proto(1, (3, 3, 32, 64), 4096, 3 * 3 * 32 * 64 * 4, 0x12345678)
    # dtype DT_FLOAT, shape {dim {size 3} dim {size 3} dim {size 32} dim {size 64}}, offset 4096, size 73728, fixed32 crc
    assert e == (b"\x08\x01" + b"\x12\x10" + b"\x12\x02\x08\x03" * 2 + b"\x12\x02\x08\x20" + b"\x12\x02\x08\x40" +
                 b"\x20\x80\x20" + b"\x28\x80\xc0\x04" + b"\x35\x78\x56\x34\x12")
    p = tfc._parse_entry(e)
    assert (p["dtype"], p["shape"], p["offset"], p["size"], p["crc32c"]) == (1, [3, 3, 32, 64], 4096, 73728, 0x12345678)
    assert tfc._parse_entry(tfc._entry_proto(9, (), 0, 8, 7))["shape"] == []        # scalar int64 (global_step)


def _hand_index(path, records):
    """An index written straight from the format description: ONE data block without prefix compression (every entry a
    restart point), an index block with one entry, an empty metaindex block."""
    def block(entries):
        body, restarts = b"", []
        for k, v in entries:
            restarts.append(len(body))
            body += tfc.put_varint(0) + tfc.put_varint(len(k)) + tfc.put_varint(len(v)) + k + v
        return body + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def trailer(b):
        return b"\x00" + struct.pack("<I", tfc.mask(tfc.crc32c(b + b"\x00")))
    data = block(records)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)            # no entries, one restart (offset 0)
    index = block([(records[-1][0] + b"\x00", tfc.put_varint(0) + tfc.put_varint(len(data)))])
    out = data + trailer(data)
    moff = len(out)
    out += meta + trailer(meta)
    ioff = len(out)
    out += index + trailer(index)
    footer = tfc.put_varint(moff) + tfc.put_varint(len(meta)) + tfc.put_varint(ioff) + tfc.put_varint(len(index))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    open(path, "wb").write(out)


def test_reads_hand_assembled_bundle(tmp_path):
    w = np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.5
    step = np.asarray(41, dtype=np.int64)
    prefix = str(tmp_path / "model.ckpt-41")
    open(prefix + ".data-00000-of-00001", "wb").write(w.tobytes() + step.tobytes())
    recs = [(b"", b"\x08\x01\x1a\x02\x08\x01"),
            (b"a/W", tfc._entry_proto(1, w.shape, 0, w.nbytes, tfc.mask(tfc.crc32c(w.tobytes())))),
            (b"global_step", tfc._entry_proto(9, (), w.nbytes, 8, tfc.mask(tfc.crc32c(step.tobytes()))))]
    _hand_index(prefix + ".index", recs)
    got = tfc.read(prefix)
    assert set(got) == {"a/W", "global_step"} and got["a/W"].dtype == np.float32 and got["global_step"].dtype == np.int64
    np.testing.assert_array_equal(got["a/W"], w)
    assert int(got["global_step"]) == 41
    assert tfc.list_variables(prefix) == {"a/W": (np.dtype(np.float32), (2, 3, 4)), "global_step": (np.dtype(np.int64), ())}
    # a flipped data byte is caught by the tensor checksum, a flipped index byte by the block checksum
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read(prefix)
    assert tfc.read(prefix, verify=False)["a/W"].shape == (2, 3, 4)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read(prefix)
    idx[-1] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="magic"):
        tfc.read(prefix)


def test_write_read_round_trip_many_blocks(tmp_path, monkeypatch):
    """700 variables with the model's name structure: prefix-compressed keys, restart points, several data blocks (block size
    lowered), every dtype the model stores."""
    monkeypatch.setattr(tfc, "BLOCK_SIZE", 2048)
    rng = np.random.default_rng(3)
    tensors = {}
    for net in ("posterior", "prior", "likelihood"):
        for i in range(60):
            base = "%s/z%d_pre_%d" % (net, i % 7, i)
            tensors[base + "/W"] = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
            tensors[base + "/W/Adam"] = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
            tensors[base + "/W/Adam_1"] = rng.random((3, 3, 4, 8)).astype(np.float32)
            tensors[base + "/batch_norm/BatchNorm/moving_mean"] = rng.standard_normal(8).astype(np.float32)
    tensors["global_step"] = np.asarray(123456789012, dtype=np.int64)
    tensors["beta1_power"] = np.asarray(0.9 ** 5, dtype=np.float32)
    tensors["flags"] = np.array([True, False, True])
    prefix = str(tmp_path / "sub" / "model_best_dice.ckpt-7")
    tfc.write(prefix, tensors)
    got = tfc.read(prefix)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    sub = tfc.read(prefix, names={"global_step", "prior/z3_pre_3/W"})
    assert sorted(sub) == ["global_step", "prior/z3_pre_3/W"]
    # the data file is the tensors back to back in key order
    order = sorted(tensors, key=lambda k: k.encode())
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(tensors[k].nbytes for k in order)
    first = np.fromfile(prefix + ".data-00000-of-00001", dtype=tensors[order[0]].dtype, count=tensors[order[0]].size)
    np.testing.assert_array_equal(first, tensors[order[0]].reshape(-1))
    from phiseg_code_amd.tfwrapper import utils as tfutils
    assert tfutils.get_latest_model_checkpoint_path(str(tmp_path / "sub"), "model_best_dice.ckpt") == prefix
    assert set(tfutils.get_checkpoint_weights(prefix)) == set(tensors)
