"""The oracle's decoder with a preset dictionary (&Reader::new_dict, inflate.mbt:315-317;
DictDecoder::new, dict-decoder.mbt:40-60), pinned against zlib's raw inflate with the same zdict: the
reference holds no fixture for this path (deflate_test.mbt:25-35 only covers the encoder's odd
new_dict), zlib is the independent decoder."""
import zlib

import numpy as np
import pytest

from util import flate


def zdeflate(data, zdict, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, zdict)
    return co.compress(data) + co.flush()


def words(seed, n):
    return flate.synth("text", 1, n, seed=seed).tobytes()


@pytest.mark.parametrize("dlen", [1, 7, 258, 4096, 32767, 32768, 32769, 50000])
def test_dictionary_streams_decode_as_zlib_decodes_them(oracle, dlen):
    zdict = words(11, dlen)
    # data that shares its vocabulary with the dictionary, so that copies reach into it
    data = zdict[-min(dlen, 3000):] + words(12, 20000) + zdict[:min(dlen, 5000)]
    comp = zdeflate(data, zdict)
    assert zlib.decompressobj(-15, zdict).decompress(comp) == data
    assert len(comp) < len(zdeflate(data, b"\0")) or dlen < 258
    rc, got, used, eo = oracle.inflate(comp, len(data) + 8, full=True, zdict=zdict)
    assert (rc, got, used) == (0, data, len(comp))


def test_without_the_dictionary_the_first_far_copy_is_corrupt(oracle):
    zdict = words(21, 8192)
    data = zdict[1000:1400] + b"tail"
    comp = zdeflate(data, zdict)
    rc, got, used, eo = oracle.inflate(comp, 4096, full=True)
    assert rc == oracle.E_CORRUPT and got == b"" and eo > 0
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(comp)
    # a dictionary that is too short: the copy's distance exceeds dictionary + output (:677-680)
    rc2, got2, _, eo2 = oracle.inflate(comp, 4096, full=True, zdict=zdict[-1000:])
    assert rc2 == oracle.E_CORRUPT and eo2 == eo
    # long enough: the suffix the copy needs is what counts (DictDecoder keeps the END of the dictionary)
    rc3, got3, _, _ = oracle.inflate(comp, 4096, full=True, zdict=b"x" * 5000 + zdict)
    assert rc3 == 0 and got3 == data


def test_copy_that_runs_from_the_dictionary_into_the_output(oracle):
    # distance 4 with 3 bytes of output: one byte from the dictionary, then the output repeats
    zdict = b"....abcdeZ"
    data = b"xyz" + b"Zxyz" * 40
    comp = zdeflate(data, zdict, 9)
    assert zlib.decompressobj(-15, zdict).decompress(comp) == data
    rc, got, _, _ = oracle.inflate(comp, 1024, full=True, zdict=zdict)
    assert rc == 0 and got == data


def test_our_own_streams_ignore_a_dictionary(oracle):
    # the encoder never refers to bytes it has not written: a dictionary changes nothing
    data = np.frombuffer(words(5, 70000), dtype=np.uint8)
    comp = oracle.deflate(data)
    assert oracle.inflate(comp, 70000, zdict=b"some dictionary") == data.tobytes()
