"""moonbit-flate_amd: MI355X-native batch engine for the deflate-fast hot path of
gmlewis/moonbit-flate (LZ77 match finder -> dynamic-Huffman bit writer, batch inflate).

The hyphen in the directory name means: importlib.import_module("moonbit-flate_amd").
"""
from .engine import (FlateEngine, FlateError, StreamWriter, build_id, deflate_bound, synth, uniform_offsets, lz_chunks,
                     tokens_from_matches, SYNTH_KINDS, SEED_TEXT, SEED_RAND, STAGES)
from . import build as _build


def source_hash():
    """Hash of the library's sources as they are on disk now (compare with build_id())."""
    return _build.source_hash()


def id_component(build_id, group):
    """One kernel family's hash ("lz77", "huff", "inflate", "all") out of a build id string."""
    return _build.id_component(build_id, group)


def build(force=False, verbose=False):
    """Compile libflate_hip.so for gfx950 (hipcc)."""
    return _build.build(force=force, verbose=verbose)
