"""Oracle Philox4x32-10 against the Random123 known-answer vectors."""
import numpy as np

from oracle import philox


def _kat(ctr, key):
    return [int(v) for v in philox.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))]


def test_random123_kats():
    assert _kat([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert _kat([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _kat([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_normal_moments_and_sharding_invariance():
    a = philox.normal(42, 3, 5, 8, 1000, dtype=np.float64)
    assert abs(a.mean()) < 0.05 and abs(a.std() - 1.0) < 0.05
    b = philox.normal(42, 3, 5, 4, 1000, sample_offset=4, dtype=np.float64)
    np.testing.assert_array_equal(a[4:], b)     # rank 1 of 2 sees the same noise as rows 4..7 of one rank
