"""ctypes loader of libflate_hip.so.  There is no fallback: a missing library is an error."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FLATE_HIP_LIB: developer override to A/B an experimental build of the same library
LIB_PATH = os.environ.get("FLATE_HIP_LIB") or os.path.join(HERE, "lib", "libflate_hip.so")

# every symbol include/flate_hip.h declares
EXPORTS = [
    "flate_hip_init", "flate_hip_destroy", "flate_hip_set_stream", "flate_hip_set_option",
    "flate_hip_strerror",
    "flate_hip_last_hip_error", "flate_hip_deflate_bound", "flate_hip_deflate_fast_batch",
    "flate_hip_lz77_matches", "flate_hip_inflate_batch", "flate_hip_deflate_fast_spliced",
    "flate_hip_inflate_spliced", "flate_hip_set_profiling", "flate_hip_last_resident_share",
    "flate_hip_last_timing", "flate_hip_stage_name", "flate_hip_synth_fill", "flate_hip_build_id",
    "flate_hip_comm_unique_id", "flate_hip_comm_init", "flate_hip_comm_wrap", "flate_hip_comm_destroy",
    "flate_hip_comm_plan", "flate_hip_comm_set_plan", "flate_hip_gather_layout",
    "flate_hip_gather_compressed", "flate_hip_gather_begin", "flate_hip_gather_end",
    "flate_hip_stream_open", "flate_hip_stream_bound", "flate_hip_stream_write", "flate_hip_stream_free",
    "flate_hip_host_register", "flate_hip_host_unregister", "flate_hip_host_alloc", "flate_hip_host_free",
    "flate_hip_inflate_stream_open", "flate_hip_inflate_stream_read", "flate_hip_inflate_stream_free",
    "flate_hip_inflate_stream_reset", "flate_hip_checksum_batch",
]

_lib = None


class FlateLibraryMissing(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlateLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (load torch's HIP runtime first so both share one libamdhip64)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.flate_hip_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.flate_hip_init.restype = C.c_int
    L.flate_hip_destroy.argtypes = [vp]
    L.flate_hip_destroy.restype = None
    L.flate_hip_set_stream.argtypes = [vp, vp]
    L.flate_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.flate_hip_strerror.argtypes = [C.c_int]
    L.flate_hip_strerror.restype = C.c_char_p
    L.flate_hip_last_hip_error.argtypes = [vp]
    L.flate_hip_last_hip_error.restype = C.c_char_p
    L.flate_hip_deflate_bound.argtypes = [C.c_size_t]
    L.flate_hip_deflate_bound.restype = C.c_size_t
    L.flate_hip_deflate_fast_batch.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, C.c_uint32]
    L.flate_hip_deflate_fast_batch.restype = C.c_int
    L.flate_hip_lz77_matches.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_uint32), u64p, vp, vp, vp]
    L.flate_hip_lz77_matches.restype = C.c_int
    L.flate_hip_inflate_batch.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32]
    L.flate_hip_inflate_batch.restype = C.c_int
    L.flate_hip_deflate_fast_spliced.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, C.c_uint32]
    L.flate_hip_deflate_fast_spliced.restype = C.c_int
    L.flate_hip_inflate_spliced.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32]
    L.flate_hip_inflate_spliced.restype = C.c_int
    L.flate_hip_set_profiling.argtypes = [vp, C.c_int]
    L.flate_hip_last_resident_share.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.flate_hip_last_resident_share.restype = C.c_int
    L.flate_hip_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.flate_hip_stage_name.argtypes = [C.c_int]
    L.flate_hip_stage_name.restype = C.c_char_p
    L.flate_hip_synth_fill.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64,
                                       vp, C.c_int]
    L.flate_hip_synth_fill.restype = C.c_int
    L.flate_hip_build_id.argtypes = []
    L.flate_hip_build_id.restype = C.c_char_p
    L.flate_hip_comm_unique_id.argtypes = [vp]
    L.flate_hip_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.flate_hip_comm_wrap.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.flate_hip_comm_destroy.argtypes = [vp]
    L.flate_hip_comm_destroy.restype = None
    L.flate_hip_comm_plan.argtypes = [vp, u64p, C.POINTER(C.c_uint32)]
    L.flate_hip_comm_set_plan.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.flate_hip_gather_layout.argtypes = [C.c_uint32, vp, C.c_uint64, C.c_uint32, u64p, vp, u64p]
    L.flate_hip_gather_compressed.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint64,
                                              vp, vp, C.c_uint64, u64p, C.c_uint32]
    L.flate_hip_gather_begin.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint64]
    L.flate_hip_gather_end.argtypes = [vp, vp, vp, C.c_uint64, u64p]
    L.flate_hip_stream_open.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.flate_hip_stream_bound.argtypes = [C.c_size_t]
    L.flate_hip_stream_bound.restype = C.c_size_t
    L.flate_hip_stream_write.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, u64p]
    L.flate_hip_stream_free.argtypes = [vp]
    L.flate_hip_stream_free.restype = None
    L.flate_hip_inflate_stream_open.argtypes = [vp, C.POINTER(vp)]
    L.flate_hip_inflate_stream_read.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, u64p, u64p,
                                                C.POINTER(C.c_int64)]
    L.flate_hip_inflate_stream_reset.argtypes = [vp, vp, C.c_uint64]
    if os.environ.get("FLATE_HIP_LIB") is None or hasattr(L, "flate_hip_checksum_batch"):
        # (a developer's A/B build of an older tree may lack the newest entry point; the product library may not)
        L.flate_hip_checksum_batch.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32]
    L.flate_hip_inflate_stream_free.argtypes = [vp]
    L.flate_hip_inflate_stream_free.restype = None
    L.flate_hip_host_register.argtypes = [vp, vp, C.c_size_t]
    L.flate_hip_host_unregister.argtypes = [vp, vp]
    L.flate_hip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.flate_hip_host_free.argtypes = [vp, vp]
    _lib = L
    return L
