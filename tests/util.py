"""Shared helpers for the parity tests."""
import importlib
import zlib

import numpy as np

flate = importlib.import_module("moonbit-flate_amd")


def raw_inflate(b):
    d = zlib.decompressobj(-15)
    out = d.decompress(bytes(b))
    assert d.eof and d.unused_data == b""
    return out


def make_streams(specs, seed=1234):
    """specs: list of (kind, length).  Returns (data uint8[], in_off uint64[n+1])."""
    rng = np.random.default_rng(seed)
    parts = []
    for i, (kind, n) in enumerate(specs):
        if kind == "ramp":
            a = (np.arange(n) & 127).astype(np.uint8)
        elif kind == "zero":
            a = np.zeros(n, dtype=np.uint8)
        elif kind == "rand":
            a = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == "text":
            a = flate.synth("text", 1, n, first_stream=1000 + i) if n else np.zeros(0, np.uint8)
        elif kind == "low":       # small alphabet, many short matches
            a = rng.integers(0, 4, n, dtype=np.uint8)
        elif kind == "period":    # periodic with noise: long matches + hash-slot reuse
            base = rng.integers(0, 256, 700, dtype=np.uint8)
            a = np.resize(base, n).copy()
            if n:
                idx = rng.integers(0, n, max(1, n // 300))
                a[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
        elif kind == "runs":      # runs of equal bytes: same-slot collisions inside a batch
            vals = rng.integers(0, 256, n // 8 + 1, dtype=np.uint8)
            reps = rng.integers(1, 40, n // 8 + 1)
            a = np.repeat(vals, reps)[:n].astype(np.uint8)
            if a.size < n:
                a = np.concatenate([a, np.zeros(n - a.size, np.uint8)])
        else:
            raise ValueError(kind)
        parts.append(a)
    lens = np.array([p.size for p in parts], dtype=np.uint64)
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    return data, off


def oracle_tokens_per_chunk(oracle, stream_bytes, compat=0):
    """Token arrays the reference's DeflateFast::encode yields for each LZ77 chunk."""
    df = oracle.DeflateFast(compat)
    out = []
    for start, n in flate.lz_chunks(len(stream_bytes)):
        out.append(df.encode(stream_bytes[start:start + n]))
    return out
