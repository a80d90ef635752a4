"""Host-side handle of the MI355X batch DEFLATE engine (thin wrapper over the C ABI).

The C ABI (include/flate_hip.h) is the product boundary; this module only marshals
numpy / torch buffers into it.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

DEVICE_PTRS = 0x1
COMPAT_GO = 0x2
LZ_SERIAL = 0x4
SIZE_ONLY = 0x8

SYNTH_RAMP, SYNTH_TEXT, SYNTH_RAND, SYNTH_ZERO = 0, 1, 2, 3
SYNTH_KINDS = {"ramp": SYNTH_RAMP, "text": SYNTH_TEXT, "rand": SYNTH_RAND, "zero": SYNTH_ZERO}
SEED_TEXT = 0x5EED0001
SEED_RAND = 0x5EED0002

STAGES = ("lz77_match", "huff_pack", "checksum", "inflate")

MAX_STORE_BLOCK_SIZE = 65535
MATCH_CAP_PER_CHUNK = 16384


class FlateError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("flate_hip error %d: %s" % (code, msg))
        self.code = code


def build_id():
    """Source hash compiled into the loaded library (see build.source_hash)."""
    return _lib.load().flate_hip_build_id().decode()


def deflate_bound(n):
    return int(_lib.load().flate_hip_deflate_bound(int(n)))


def synth(kind, n_streams, stream_len, seed=None, first_stream=0, nthreads=None):
    """Synthetic benchmark input: n_streams streams of stream_len bytes, back to back."""
    k = SYNTH_KINDS[kind] if isinstance(kind, str) else int(kind)
    if seed is None:
        seed = SEED_RAND if k == SYNTH_RAND else SEED_TEXT
    if nthreads is None:
        nthreads = min(16, os.cpu_count() or 1)
    out = np.empty(int(n_streams) * int(stream_len), dtype=np.uint8)
    rc = _lib.load().flate_hip_synth_fill(k, seed, first_stream, n_streams, stream_len,
                                          out.ctypes.data, nthreads)
    if rc != 0:
        raise FlateError(rc, "synth_fill")
    return out


def uniform_offsets(n_streams, stream_len):
    return (np.arange(n_streams + 1, dtype=np.uint64) * np.uint64(stream_len))


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _check_out(out, data, need, what):
    """A caller-supplied output buffer must be uint8, contiguous, on the same side (and device) as
    the input and hold at least `need` bytes: the kernels bound their writes by the slot table,
    not by the real size of `out`."""
    if _is_torch(data):
        import torch
        ok = _is_torch(out) and out.dtype == torch.uint8 and out.is_contiguous() and \
            out.device == data.device and out.numel() >= need
    else:
        ok = isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and \
            out.size >= need
    if not ok:
        raise FlateError(E_INVALID, "%s: out must be a contiguous uint8 buffer of >= %d bytes on the "
                                    "same device as the input" % (what, need))


class FlateEngine:
    """One engine = one flate_hip_ctx = one GPU + one HIP stream."""

    def __init__(self, device=0):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._L.flate_hip_init(int(device), C.byref(self._ctx))
        if rc != 0:
            raise FlateError(rc, self._L.flate_hip_strerror(rc).decode())
        self.device = int(device)

    def close(self):
        if self._ctx:
            self._L.flate_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._L.flate_hip_strerror(rc).decode()
            extra = self._L.flate_hip_last_hip_error(self._ctx).decode()
            raise FlateError(rc, msg + (" [" + extra + "]" if extra else ""))

    def use_stream(self, hip_stream_ptr):
        """Launch on the given hipStream_t (int address), e.g. torch.cuda.current_stream().cuda_stream."""
        self._check(self._L.flate_hip_set_stream(self._ctx, C.c_void_p(hip_stream_ptr or None)))

    def host_register(self, arr):
        """Page-lock a host buffer the caller keeps across calls (flate_hip_host_register): the copies of
        the host-pointer calls then run at the link's rate.  Returns a context manager that
        unregisters on exit (or call .close())."""
        a = np.ascontiguousarray(arr)
        assert a is arr or np.shares_memory(a, arr), "host_register needs a contiguous array"
        self._check(self._L.flate_hip_host_register(self._ctx, a.ctypes.data, a.nbytes))
        eng = self

        class _Reg:
            def __init__(self):
                self.ptr = a.ctypes.data

            def close(self):
                if self.ptr:
                    eng._check(eng._L.flate_hip_host_unregister(eng._ctx, self.ptr))
                    self.ptr = None

            def __enter__(self):
                return self

            def __exit__(self, *exc):
                self.close()

        return _Reg()

    def set_option(self, name, value):
        """Launch-geometry knobs (guest_blocks, resident_blocks, guest_min_streams)."""
        self._check(self._L.flate_hip_set_option(self._ctx, name.encode(), int(value)))

    def set_profiling(self, on=True):
        self._check(self._L.flate_hip_set_profiling(self._ctx, 1 if on else 0))

    def last_resident_share(self):
        """(streams the LDS-table blocks took, streams queued) of the last persistent match-finder launch."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.flate_hip_last_resident_share(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def last_timing(self):
        ms = (C.c_float * len(STAGES))()
        self._check(self._L.flate_hip_last_timing(self._ctx, ms, len(STAGES)))
        return {STAGES[i]: float(ms[i]) for i in range(len(STAGES))}

    @staticmethod
    def _flags(compat_go, lz_serial, device):
        return (COMPAT_GO if compat_go else 0) | (LZ_SERIAL if lz_serial else 0) | \
               (DEVICE_PTRS if device else 0)

    def deflate_batch(self, data, in_off, out=None, out_cap=None, compat_go=False, lz_serial=False):
        """Compress independent streams: stream i = data[in_off[i]:in_off[i+1]] with fresh-Writer
        semantics.  data: numpy uint8 array (host) or torch uint8 CUDA tensor (device).
        Returns (out, out_off): out has the same kind as data, out_off is numpy uint64[n+1]."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        device = _is_torch(data)
        if out_cap is None and out is None:
            lens = in_off[1:] - in_off[:-1]
            out_cap = sum(deflate_bound(int(l)) * int(c)
                          for l, c in zip(*np.unique(lens, return_counts=True)))
        if device:
            import torch
            assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous()
            if out is None:
                out = torch.empty(max(int(out_cap), 16), dtype=torch.uint8, device=data.device)
            else:
                _check_out(out, data, 1, "deflate_batch")
            cap = out.numel()
            in_ptr, out_ptr = data.data_ptr(), out.data_ptr()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            if out is None:
                out = np.empty(max(int(out_cap), 16), dtype=np.uint8)
            else:
                _check_out(out, data, 1, "deflate_batch")
            cap = out.size
            in_ptr, out_ptr = data.ctypes.data, out.ctypes.data
        rc = self._L.flate_hip_deflate_fast_batch(self._ctx, in_ptr, in_off.ctypes.data, n, out_ptr,
                                                  cap, out_off.ctypes.data,
                                                  self._flags(compat_go, lz_serial, device))
        self._check(rc)
        return out, out_off

    def checksum_batch(self, data, in_off, kind="adler32"):
        """Adler-32 (RFC 1950) or CRC-32 (RFC 1952) of every stream data[in_off[i]:in_off[i+1]] -> numpy uint32[n]
        (flate_hip_checksum_batch; numpy data = host pointers, torch CUDA tensor = device pointers)."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        out = np.zeros(max(n, 1), dtype=np.uint32)
        device = _is_torch(data)
        if device:
            ptr = data.data_ptr()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            ptr = data.ctypes.data if data.size else None
        k = {"adler32": CHECKSUM_ADLER32, "crc32": CHECKSUM_CRC32}[kind]
        self._check(self._L.flate_hip_checksum_batch(self._ctx, ptr, in_off.ctypes.data, n, k, out.ctypes.data,
                                                     DEVICE_PTRS if device else 0))
        return out[:n]

    def deflate_batch_framed(self, data, in_off, wrap, compat_go=False):
        """The streams of a batch as zlib (RFC 1950) or gzip (RFC 1952) members: the raw DEFLATE streams of
        deflate_batch between the container's header and its trailer -- Adler-32, or CRC-32 and the length,
        computed on the GPU (flate_hip_checksum_batch).  Host data; returns (bytes array, off[n+1])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        raw, roff = self.deflate_batch(data, in_off, compat_go=compat_go)
        sums = self.checksum_batch(data, in_off, "adler32" if wrap == "zlib" else "crc32")
        head = ZLIB_HEADER if wrap == "zlib" else GZIP_HEADER
        tail = 4 if wrap == "zlib" else 8
        off = roff + np.arange(n + 1, dtype=np.uint64) * np.uint64(len(head) + tail)
        out = np.empty(int(off[-1]), dtype=np.uint8)
        lens = (in_off[1:] - in_off[:-1]).astype(np.uint64)
        for i in range(n):
            o = int(off[i])
            out[o:o + len(head)] = np.frombuffer(head, np.uint8)
            o += len(head)
            k = int(roff[i + 1] - roff[i])
            out[o:o + k] = raw[int(roff[i]):int(roff[i + 1])]
            o += k
            if wrap == "zlib":
                out[o:o + 4] = np.frombuffer(int(sums[i]).to_bytes(4, "big"), np.uint8)
            else:
                out[o:o + 8] = np.frombuffer(int(sums[i]).to_bytes(4, "little") +
                                             (int(lens[i]) & 0xFFFFFFFF).to_bytes(4, "little"), np.uint8)
        return out, off

    def deflate_spliced_framed(self, data, in_off, wrap, compat_go=False):
        """The whole batch as ONE zlib stream or ONE gzip member: the spliced DEFLATE stream of deflate_spliced
        (every input stream compressed on its own, in parallel, joined at bit granularity) between the
        container's header and the checksum of ALL the input -- what `gzip -d` / zlib.decompress turn back
        into the concatenated input.  Host data; returns bytes."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        one, nbytes, _ = self.deflate_spliced(data, in_off, compat_go=compat_go)
        total = int(in_off[-1] - in_off[0])
        whole = np.array([in_off[0], in_off[-1]], dtype=np.uint64)
        s = int(self.checksum_batch(data, whole, "adler32" if wrap == "zlib" else "crc32")[0])
        if wrap == "zlib":
            return ZLIB_HEADER + bytes(one[:int(nbytes)]) + s.to_bytes(4, "big")
        return GZIP_HEADER + bytes(one[:int(nbytes)]) + s.to_bytes(4, "little") + (total & 0xFFFFFFFF).to_bytes(4, "little")

    def inflate_batch_framed(self, data, in_off, wrap, out_sizes=None):
        """The reverse: zlib or gzip members -> (out, out_off, status[n]); status -4 (FLATE_HIP_E_CORRUPT) also
        for a bad header, a checksum or (gzip) a length that does not match.  zlib members carry no size:
        out_sizes (or a size-only pass of the decoder) supplies it; gzip's ISIZE is used as the slot size."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        start = np.zeros(n + 1, dtype=np.uint64)
        want = np.zeros(n, dtype=np.uint32)
        isize = np.zeros(n, dtype=np.uint64)
        bad = np.zeros(n, dtype=bool)
        for i in range(n):
            a, b = int(in_off[i]), int(in_off[i + 1])
            m = data[a:b]
            h, t = parse_container_header(m, wrap)
            if h < 0 or b - a < h + t:
                bad[i] = True
                h = 0
            else:
                tr = bytes(m[len(m) - t:])
                want[i] = int.from_bytes(tr[:4], "big" if wrap == "zlib" else "little")
                if wrap == "gzip":
                    isize[i] = int.from_bytes(tr[4:8], "little")
            start[i] = a + h
        start[n] = in_off[n]
        src = np.concatenate([data[:int(in_off[n])], np.zeros(8, np.uint8)])
        if out_sizes is None:
            if wrap == "gzip":
                out_sizes = isize
            else:
                out_sizes, _, _ = self.inflate_sizes(src, start)
        out, ooff, olen, status, _ = self.inflate_batch(src, start, out_sizes, check=False)
        sums = self.checksum_batch(out, ooff, "adler32" if wrap == "zlib" else "crc32") if n else np.zeros(0, np.uint32)
        status = np.array(status, dtype=np.int32)
        for i in range(n):
            if bad[i]:
                status[i] = E_CORRUPT
            elif status[i] == 0:
                # (a slot larger than the member's output: the checksum is over the bytes produced)
                if int(olen[i]) != int(ooff[i + 1] - ooff[i]):
                    s = self.checksum_batch(out[int(ooff[i]):int(ooff[i]) + int(olen[i])], [0, int(olen[i])],
                                            "adler32" if wrap == "zlib" else "crc32")[0]
                else:
                    s = sums[i]
                if int(s) != int(want[i]) or (wrap == "gzip" and (int(olen[i]) & 0xFFFFFFFF) != int(isize[i])):
                    status[i] = E_CORRUPT
        return out, ooff, olen, status

    def inflate_batch(self, data, in_off, out_sizes, out=None, check=True):
        """Decompress independent DEFLATE streams (&Reader::new + read to EOF each).
        out_sizes[i] = capacity reserved for stream i's output (its exact size if known).
        Returns (out, out_off, out_len, status, err_off); with check=True a failing stream raises."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray(out_sizes, dtype=np.uint64), out=out_off[1:])
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        err_off = np.full(n, -1, dtype=np.int64)
        device = _is_torch(data)
        total = max(int(out_off[-1]), 16)
        if device:
            import torch
            assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous()
            if out is None:
                out = torch.empty(total, dtype=torch.uint8, device=data.device)
            else:
                _check_out(out, data, int(out_off[-1]), "inflate")
            in_ptr, out_ptr = data.data_ptr(), out.data_ptr()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            if out is None:
                out = np.zeros(total, dtype=np.uint8)
            else:
                _check_out(out, data, int(out_off[-1]), "inflate")
            in_ptr, out_ptr = data.ctypes.data, out.ctypes.data
        rc = self._L.flate_hip_inflate_batch(self._ctx, in_ptr, in_off.ctypes.data, n, out_ptr,
                                             out_off.ctypes.data, out_len.ctypes.data,
                                             status.ctypes.data, err_off.ctypes.data,
                                             DEVICE_PTRS if device else 0)
        if rc != 0 and (check or rc not in (E_OUT_TOO_SMALL, E_CORRUPT, E_UNEXPECTED_EOF)):
            self._check(rc)
        return out, out_off, out_len, status, err_off

    def inflate_sizes(self, data, in_off):
        """Decode without storing (FLATE_HIP_SIZE_ONLY): returns (out_len, status, err_off) -- the size
        every stream inflates to (up to its error, if any).  Host or device input."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        err_off = np.full(max(n, 1), -1, dtype=np.int64)
        device = _is_torch(data)
        if not device:
            data = np.ascontiguousarray(data, dtype=np.uint8)
        in_ptr = data.data_ptr() if device else data.ctypes.data
        rc = self._L.flate_hip_inflate_batch(self._ctx, in_ptr, in_off.ctypes.data, n, None, None,
                                             out_len.ctypes.data, status.ctypes.data, err_off.ctypes.data,
                                             SIZE_ONLY | (DEVICE_PTRS if device else 0))
        if rc != 0 and rc not in (E_OUT_TOO_SMALL, E_CORRUPT, E_UNEXPECTED_EOF):
            self._check(rc)
        return out_len[:n], status[:n], err_off[:n]

    def open_inflate_stream(self, zdict=None):
        """One long DEFLATE stream decoded in pieces (see StreamReader); zdict: a preset dictionary
        (&Reader::new_dict, inflate.mbt:315-317)."""
        return StreamReader(self, zdict)

    def open_stream(self, compat_go=False):
        """One stream written in pieces (flate_hip_stream_*): see StreamWriter."""
        return StreamWriter(self, compat_go)

    def deflate_spliced(self, data, in_off, out=None, compat_go=False):
        """Compress the streams as deflate_batch does, but into ONE legal DEFLATE stream that
        inflates to the concatenation of the inputs.
        Returns (out, out_len, bit_off[N+1]): bit_off[i] = bit position of stream i's first block."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        lens = in_off[1:] - in_off[:-1]
        cap = int(sum(deflate_bound(int(l)) * int(c) for l, c in zip(*np.unique(lens, return_counts=True)))) + 16
        bit_off = np.zeros(n + 1, dtype=np.uint64)
        out_len = C.c_uint64(0)
        device = _is_torch(data)
        if device:
            import torch
            assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous()
            if out is None:
                out = torch.empty(cap, dtype=torch.uint8, device=data.device)
            else:
                _check_out(out, data, 8, "deflate_spliced")
            in_ptr, out_ptr, out_cap = data.data_ptr(), out.data_ptr(), out.numel()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            if out is None:
                out = np.zeros(cap, dtype=np.uint8)
            else:
                _check_out(out, data, 8, "deflate_spliced")
            in_ptr, out_ptr, out_cap = data.ctypes.data, out.ctypes.data, out.size
        self._check(self._L.flate_hip_deflate_fast_spliced(
            self._ctx, in_ptr, in_off.ctypes.data, n, out_ptr, out_cap, C.byref(out_len),
            bit_off.ctypes.data, self._flags(compat_go, False, device)))
        return out, int(out_len.value), bit_off

    def inflate_spliced(self, data, nbytes, bit_off, out_sizes, out=None, check=True):
        """Decompress ONE spliced DEFLATE stream data[:nbytes] in parallel from its index:
        piece i starts at bit bit_off[i].  Returns (out, out_off, out_len, status, err_off)."""
        bit_off = np.ascontiguousarray(bit_off, dtype=np.uint64)
        n = bit_off.size - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray(out_sizes, dtype=np.uint64), out=out_off[1:])
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        err_off = np.full(max(n, 1), -1, dtype=np.int64)
        device = _is_torch(data)
        total = max(int(out_off[-1]), 16)
        if device:
            import torch
            assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous()
            if out is None:
                out = torch.empty(total, dtype=torch.uint8, device=data.device)
            else:
                _check_out(out, data, int(out_off[-1]), "inflate")
            in_ptr, out_ptr = data.data_ptr(), out.data_ptr()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            if out is None:
                out = np.zeros(total, dtype=np.uint8)
            else:
                _check_out(out, data, int(out_off[-1]), "inflate")
            in_ptr, out_ptr = data.ctypes.data, out.ctypes.data
        rc = self._L.flate_hip_inflate_spliced(self._ctx, in_ptr, int(nbytes), bit_off.ctypes.data, n,
                                               out_ptr, out_off.ctypes.data, out_len.ctypes.data,
                                               status.ctypes.data, err_off.ctypes.data,
                                               DEVICE_PTRS if device else 0)
        if rc != 0 and (check or rc not in (E_OUT_TOO_SMALL, E_CORRUPT, E_UNEXPECTED_EOF)):
            self._check(rc)
        return out, out_off, out_len[:n], status[:n], err_off[:n]

    def lz77_matches(self, data, in_off, compat_go=False, lz_serial=False):
        """Match finder only.  Returns a list over LZ77 chunks (stream order) of
        (pos uint32[], tok uint32[]) and the per-stream chunk counts."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n_chunks, cap = C.c_uint32(0), C.c_uint64(0)
        flags = self._flags(compat_go, lz_serial, False)
        self._check(self._L.flate_hip_lz77_matches(self._ctx, data.ctypes.data, in_off.ctypes.data, n,
                                                   flags, C.byref(n_chunks), C.byref(cap),
                                                   None, None, None))
        nc = n_chunks.value
        nmatch = np.zeros(max(nc, 1), dtype=np.uint32)
        rec_off = np.zeros(nc + 1, dtype=np.uint64)
        recs = np.zeros((max(cap.value, 1), 2), dtype=np.uint32)
        self._check(self._L.flate_hip_lz77_matches(self._ctx, data.ctypes.data, in_off.ctypes.data, n,
                                                   flags, C.byref(n_chunks), C.byref(cap),
                                                   nmatch.ctypes.data, rec_off.ctypes.data,
                                                   recs.ctypes.data))
        out = []
        for c in range(nc):
            r = recs[int(rec_off[c]):int(rec_off[c]) + int(nmatch[c])]
            out.append((r[:, 0].copy(), r[:, 1].copy()))
        return out


class StreamWriter:
    """Writer::write as the reference behaves for ONE long stream: every write() of whole
    65535-byte windows returns the compressed bytes that are complete so far; close(tail) ends the
    stream.  The concatenation equals deflate_batch of the whole stream, bit for bit."""
    WINDOW = MAX_STORE_BLOCK_SIZE

    def __init__(self, eng, compat_go=False):
        self._eng, self._L = eng, eng._L
        self._st = C.c_void_p()
        eng._check(self._L.flate_hip_stream_open(eng._ctx, COMPAT_GO if compat_go else 0, C.byref(self._st)))

    def _write(self, piece, final):
        piece = np.ascontiguousarray(piece, dtype=np.uint8)
        cap = int(self._L.flate_hip_stream_bound(piece.size))
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        rc = self._L.flate_hip_stream_write(self._st, piece.ctypes.data if piece.size else None, piece.size,
                                            1 if final else 0, out.ctypes.data, cap, C.byref(n))
        self._eng._check(rc)
        return out[:int(n.value)]

    def write(self, piece):
        """piece: a multiple of 65535 bytes.  Returns the output bytes finished by it."""
        return self._write(piece, False)

    def close(self, tail=b""):
        """The rest of the stream (any length) and Writer::close.  Returns the last output bytes."""
        tail = np.frombuffer(bytes(tail), dtype=np.uint8) if not isinstance(tail, np.ndarray) else tail
        out = self._write(tail, True)
        self.free()
        return out

    def free(self):
        if self._st:
            self._L.flate_hip_stream_free(self._st)
            self._st = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class StreamReader:
    """Decompressor::read as the reference behaves for ONE long stream (flate_hip_inflate_stream_*): the
    caller feeds pieces of the compressed stream and takes pieces of output; the decoder's state rests
    on the device in between.  `feed(piece, final, room)` returns (out bytes, status): status 0 = go on,
    1 = end of stream, < 0 = the stream's error (err_off holds corrupt_input_error's offset)."""

    def __init__(self, eng, zdict=None):
        self._eng, self._L = eng, eng._L
        self._st = C.c_void_p()
        eng._check(self._L.flate_hip_inflate_stream_open(eng._ctx, C.byref(self._st)))
        self._rest = np.zeros(0, dtype=np.uint8)  # bytes a call reported as unused
        self.err_off = -1
        self.total_in = 0
        if zdict is not None and len(zdict):
            self.reset(zdict)

    def reset(self, zdict=None):
        """Decompressor::reset(r, dict) (inflate.mbt:862-884): a fresh decoder on the same handle, with the
        last 32768 bytes of zdict as history that has already been read."""
        d = np.ascontiguousarray(np.frombuffer(bytes(zdict), dtype=np.uint8) if zdict is not None and not isinstance(zdict, np.ndarray)
                                 else (zdict if zdict is not None else np.zeros(0, np.uint8)), dtype=np.uint8)
        self._eng._check(self._L.flate_hip_inflate_stream_reset(self._st, d.ctypes.data if d.size else None, d.size))
        self._rest = np.zeros(0, dtype=np.uint8)
        self.err_off = -1
        self.total_in = 0

    def feed(self, piece, final=False, room=1 << 20):
        piece = np.ascontiguousarray(np.frombuffer(bytes(piece), dtype=np.uint8) if not isinstance(piece, np.ndarray) else piece,
                                     dtype=np.uint8)
        buf = np.concatenate([self._rest, piece]) if self._rest.size else piece
        out = np.empty(max(int(room), 1), dtype=np.uint8)
        used, n, eo = C.c_uint64(0), C.c_uint64(0), C.c_int64(-1)
        rc = self._L.flate_hip_inflate_stream_read(self._st, buf.ctypes.data if buf.size else None, buf.size,
                                                   1 if final else 0, out.ctypes.data, int(room), C.byref(used),
                                                   C.byref(n), C.byref(eo))
        if rc not in (0, 1, E_CORRUPT, E_UNEXPECTED_EOF):
            self._eng._check(rc)
        self._rest = buf[int(used.value):].copy()
        self.total_in += int(used.value)
        self.err_off = int(eo.value)
        return out[:int(n.value)], rc

    @property
    def pending_input(self):
        return int(self._rest.size)

    def free(self):
        if self._st:
            self._L.flate_hip_inflate_stream_free(self._st)
            self._st = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


CHECKSUM_ADLER32, CHECKSUM_CRC32 = 1, 2
ZLIB_HEADER = bytes([0x78, 0x01])  # CM = 8, CINFO = 7 (32 KiB window), FLEVEL = 0 (fastest), FCHECK (RFC 1950 2.2)
GZIP_HEADER = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 255])  # no name / time, XFL = 4 (fastest), OS unknown (RFC 1952 2.3)


def parse_container_header(m, wrap):
    """(header bytes, trailer bytes) of one zlib / gzip member, or (-1, 0) if its header is not valid
    (RFC 1950 2.2: CM = 8, window <= 32 KiB, FCHECK, no preset dictionary; RFC 1952 2.3: magic, CM = 8,
    reserved flag bits zero, the optional fields skipped)."""
    m = bytes(m[:min(len(m), 70000)])
    if wrap == "zlib":
        if len(m) < 2 or (m[0] & 15) != 8 or (m[0] >> 4) > 7 or ((m[0] << 8) | m[1]) % 31 or (m[1] & 0x20):
            return -1, 0
        return 2, 4
    if len(m) < 10 or m[0] != 0x1f or m[1] != 0x8b or m[2] != 8 or (m[3] & 0xe0):
        return -1, 0
    flg, p = m[3], 10
    if flg & 4:  # FEXTRA
        if len(m) < p + 2:
            return -1, 0
        p += 2 + int.from_bytes(m[p:p + 2], "little")
    for bit in (8, 16):  # FNAME, FCOMMENT: zero-terminated
        if flg & bit:
            z = m.find(b"\0", p)
            if z < 0:
                return -1, 0
            p = z + 1
    if flg & 2:  # FHCRC
        p += 2
    return (p, 8) if p <= len(m) else (-1, 0)


# status codes of inflate_batch (include/flate_hip.h)
E_INVALID, E_OUT_TOO_SMALL, E_CORRUPT, E_UNEXPECTED_EOF, E_INTERNAL = -1, -2, -4, -7, -8


def lz_chunks(stream_len):
    """(start, length) of the LZ77 chunks of one stream (Compressor::enc_speed policy)."""
    full, r = divmod(int(stream_len), MAX_STORE_BLOCK_SIZE)
    ch = [(i * MAX_STORE_BLOCK_SIZE, MAX_STORE_BLOCK_SIZE) for i in range(full)]
    if r >= 128:
        ch.append((full * MAX_STORE_BLOCK_SIZE, r))
    return ch


def tokens_from_matches(chunk_bytes, pos, tok):
    """Expand match records into the reference's token array (token.mbt:69,76)."""
    src = np.frombuffer(bytes(chunk_bytes), dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) \
        else chunk_bytes
    n = src.size
    pos = pos.astype(np.int64)
    lens = ((tok >> 22) & 0xFF).astype(np.int64) + 3
    covered = np.zeros(n + 1, dtype=np.int64)
    np.add.at(covered, pos, 1)
    np.add.at(covered, np.minimum(pos + lens, n), -1)
    inside = np.cumsum(covered[:n]) > 0
    is_start = np.zeros(n, dtype=bool)
    is_start[pos] = True
    keep = (~inside) | is_start
    vals = src.astype(np.uint32)
    vals[pos] = tok
    return vals[keep]
