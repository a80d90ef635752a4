"""ctypes binding of the CPU oracle (oracle/libflate_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libflate_oracle.so")

COMPAT_MOONBIT = 0
COMPAT_GO = 1

OK = 0
E_OUT_TOO_SMALL = -1
E_CORRUPT = -2
E_UNEXPECTED_EOF = -3
E_INTERNAL = -4

MAX_STORE_BLOCK_SIZE = 65535


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    # (always through make: it rebuilds only when a source is newer than the library)
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["libflate_oracle.so"])
    return _LIB_PATH


class BlockInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_len", C.c_int), ("ntokens", C.c_int),
                ("bit_start", C.c_longlong)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, u32p, i32p, u64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_uint64))
        L.orc_deflate_stream.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int,
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int]
        L.orc_deflate_stream.restype = C.c_int
        L.orc_inflate_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_longlong)]
        L.orc_inflate_stream.restype = C.c_int
        L.orc_inflate_stream_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_longlong)]
        L.orc_inflate_stream_dict.restype = C.c_int
        L.orc_adler32.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_adler32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_crc32.restype = C.c_uint32
        L.orc_frame_overhead.argtypes = [C.c_int]
        L.orc_frame_overhead.restype = C.c_size_t
        L.orc_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_frame.restype = C.c_size_t
        L.orc_deflate_bound.argtypes = [C.c_size_t]
        L.orc_deflate_bound.restype = C.c_size_t
        L.orc_df_new.argtypes = [C.c_int]
        L.orc_df_new.restype = C.c_void_p
        L.orc_df_free.argtypes = [C.c_void_p]
        L.orc_df_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_df_encode.restype = C.c_int
        L.orc_df_reset.argtypes = [C.c_void_p]
        L.orc_df_cur.argtypes = [C.c_void_p]
        L.orc_df_cur.restype = C.c_int32
        L.orc_test_set_buffer_reset.argtypes = [C.c_int32]
        L.orc_huffman_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_last_blocks.argtypes = [C.POINTER(BlockInfo), C.c_int]
        L.orc_last_blocks.restype = C.c_int
        L.orc_deflate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_deflate_batch.restype = C.c_int
        L.orc_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int]
        L.orc_inflate_batch.restype = C.c_int
        L.orc_deflate_spliced.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p, C.c_int]
        L.orc_deflate_spliced.restype = C.c_int
        for name in ("orc_token_offset", "orc_token_length", "orc_token_literal",
                     "orc_reverse16", "orc_hash"):
            getattr(L, name).argtypes = [C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        L.orc_reverse_bits.argtypes = [C.c_uint32, C.c_int]
        L.orc_reverse_bits.restype = C.c_uint32
        L.orc_match_token.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_match_token.restype = C.c_uint32
        L.orc_length_code.argtypes = [C.c_uint32]
        L.orc_length_code.restype = C.c_int
        L.orc_offset_code.argtypes = [C.c_uint32]
        L.orc_offset_code.restype = C.c_int
        _lib = L
    return _lib


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def deflate(data, writes=None, compat=COMPAT_MOONBIT, with_blocks=False):
    """Writer::new + write(...)* + close  ->  compressed bytes."""
    L = lib()
    src = _as_u8(data)
    n = src.size
    cap = L.orc_deflate_bound(n)
    out = np.empty(cap, dtype=np.uint8)
    out_len = C.c_size_t(0)
    if writes is None:
        sizes, nw = None, 0
    else:
        assert sum(writes) == n
        sizes = (C.c_size_t * len(writes))(*writes)
        nw = len(writes)
    rc = L.orc_deflate_stream(src.ctypes.data, n, sizes, nw, out.ctypes.data, cap,
                              C.byref(out_len), compat)
    if rc != 0:
        raise RuntimeError("oracle deflate failed: %d" % rc)
    res = out[:out_len.value].tobytes()
    if with_blocks:
        cnt = L.orc_last_blocks(None, 0)
        arr = (BlockInfo * max(cnt, 1))()
        L.orc_last_blocks(arr, cnt)
        blocks = [(arr[i].kind, arr[i].in_len, arr[i].ntokens, arr[i].bit_start)
                  for i in range(cnt)]
        return res, blocks
    return res


def inflate(data, max_out, full=False, zdict=None):
    """&Reader::new(buf) -- or &Reader::new_dict(buf, zdict) -- read to EOF -> bytes (raises on error
    unless full)."""
    L = lib()
    src = _as_u8(data)
    out = np.empty(max(max_out, 1), dtype=np.uint8)
    out_len, consumed, err_off = C.c_size_t(0), C.c_size_t(0), C.c_longlong(-1)
    d = _as_u8(zdict) if zdict is not None and len(zdict) else None
    rc = L.orc_inflate_stream_dict(src.ctypes.data, src.size, d.ctypes.data if d is not None else None,
                                   d.size if d is not None else 0, out.ctypes.data, max_out,
                                   C.byref(out_len), C.byref(consumed), C.byref(err_off))
    res = out[:out_len.value].tobytes()
    if full:
        return rc, res, consumed.value, err_off.value
    if rc != 0:
        raise RuntimeError("oracle inflate failed: rc=%d off=%d" % (rc, err_off.value))
    return res


def inflate_batch(in_buf, in_off, out_sizes, nthreads=1):
    """Independent DEFLATE streams -> (out_buf, out_off[N+1], out_len[N], status[N])."""
    L = lib()
    src = _as_u8(in_buf)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    n = in_off.size - 1
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(np.asarray(out_sizes, dtype=np.uint64), out=out_off[1:])
    out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    L.orc_inflate_batch(src.ctypes.data, in_off.ctypes.data, n, out.ctypes.data, out_off.ctypes.data,
                        out_len.ctypes.data, status.ctypes.data, nthreads)
    return out, out_off, out_len, status


def deflate_spliced(in_buf, in_off, compat=COMPAT_MOONBIT):
    """The streams of a batch as ONE legal DEFLATE stream -> (bytes, bit_off[N+1])."""
    L = lib()
    src = _as_u8(in_buf)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    n = in_off.size - 1
    cap = sum(L.orc_deflate_bound(int(in_off[i + 1] - in_off[i])) for i in range(n)) + 16
    out = np.empty(cap, dtype=np.uint8)
    bit_off = np.zeros(n + 1, dtype=np.uint64)
    out_len = C.c_size_t(0)
    rc = L.orc_deflate_spliced(src.ctypes.data, in_off.ctypes.data, n, out.ctypes.data, cap,
                               C.byref(out_len), bit_off.ctypes.data, compat)
    if rc != 0:
        raise RuntimeError("oracle spliced deflate failed: %d" % rc)
    return out[:out_len.value].tobytes(), bit_off


def set_buffer_reset(v):
    """Test hook: buffer_reset (deflate-fast.mbt:55) of every encoder of this process; 0 = the real one."""
    lib().orc_test_set_buffer_reset(int(v))


class DeflateFast:
    """Stateful DeflateFast (deflate-fast.mbt:104) for window-by-window token parity."""

    def __init__(self, compat=COMPAT_MOONBIT):
        self._h = lib().orc_df_new(compat)

    def encode(self, window):
        src = _as_u8(window)
        toks = np.empty(src.size + 1, dtype=np.uint32)
        n = lib().orc_df_encode(self._h, toks.ctypes.data, 0, src.ctypes.data, src.size)
        return toks[:n].copy()

    def reset(self):
        lib().orc_df_reset(self._h)

    @property
    def cur(self):
        return lib().orc_df_cur(self._h)

    def __del__(self):
        try:
            lib().orc_df_free(self._h)
        except Exception:
            pass


def huffman_generate(freq, max_bits):
    f = np.ascontiguousarray(freq, dtype=np.int32)
    codes = np.zeros(f.size, dtype=np.uint32)
    lens = np.zeros(f.size, dtype=np.uint32)
    lib().orc_huffman_generate(f.ctypes.data, f.size, max_bits, codes.ctypes.data,
                               lens.ctypes.data)
    return codes, lens


def deflate_batch(in_buf, in_off, compat=COMPAT_MOONBIT, nthreads=1):
    """Independent streams -> (out_buf, out_off[N+1] slot starts, out_len[N])."""
    L = lib()
    src = _as_u8(in_buf)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    n = in_off.size - 1
    lens = (in_off[1:] - in_off[:-1]).astype(np.uint64)
    bounds = np.array([L.orc_deflate_bound(int(x)) for x in np.unique(lens)], dtype=np.uint64)
    bmap = dict(zip(np.unique(lens).tolist(), bounds.tolist()))
    slot = np.array([bmap[int(x)] for x in lens], dtype=np.uint64)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(slot, out=out_off[1:])
    out = np.empty(int(out_off[-1]), dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint64)
    rc = L.orc_deflate_batch(src.ctypes.data, in_off.ctypes.data, n, out.ctypes.data,
                             out_off.ctypes.data, out_len.ctypes.data, compat, nthreads)
    if rc != 0:
        raise RuntimeError("oracle batch deflate failed: %d" % rc)
    return out, out_off, out_len


FRAME_RAW, FRAME_ZLIB, FRAME_GZIP = 0, 1, 2


def adler32(data):
    a = _as_u8(data)
    return int(lib().orc_adler32(a.ctypes.data if a.size else None, a.size))


def crc32(data):
    a = _as_u8(data)
    return int(lib().orc_crc32(a.ctypes.data if a.size else None, a.size))


def frame(kind, raw, data):
    """The raw DEFLATE stream `raw` of `data` inside a zlib (RFC 1950) or gzip (RFC 1952) container."""
    L = lib()
    r, d = _as_u8(raw), _as_u8(data)
    out = np.empty(r.size + L.orc_frame_overhead(kind) + 1, dtype=np.uint8)
    n = L.orc_frame(kind, r.ctypes.data if r.size else None, r.size, d.ctypes.data if d.size else None, d.size,
                    out.ctypes.data)
    return out[:n].tobytes()

