"""Builds libflate_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libflate_hip.so")

SOURCES = ["lz77_kernels.hip",  "huff_pack_kernels.hip", "compact_kernels.hip",
           "inflate_kernels.hip", "splice_kernels.hip", "flate_api.hip", "gather.hip", "checksum.hip", "synth.cpp"]
HEADERS = ["flate_common.h", "flate_kernels.h", "lz77_device.h", "inflate_spec_kernel.inc", "inflate_stream_kernel.inc",
           os.path.join(ROOT, "include", "flate_hip.h")]


GROUPS = {  # which sources a kernel family's measurements depend on (besides the shared headers / ABI)
    "lz77": ["lz77_kernels.hip", "lz77_device.h"],
    "huff": ["huff_pack_kernels.hip", "compact_kernels.hip", "splice_kernels.hip"],
    "inflate": ["inflate_kernels.hip", "inflate_spec_kernel.inc", "inflate_stream_kernel.inc"],
    "checksum": ["checksum.hip"],
}
SHARED = ["flate_common.h", "flate_kernels.h", "flate_api.hip"]


def _hash_files(names):
    import hashlib
    h = hashlib.sha256()
    for n in sorted(names):
        f = n if os.path.isabs(n) else os.path.join(CSRC, n)
        if os.path.exists(f):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def source_hash():
    """'all:<h>;lz77:<h>;huff:<h>;inflate:<h>' -- hashes over the sources the library is built from:
    everything, and per kernel family (its own files + the shared headers and the ABI file).
    Compiled into the library (flate_hip_build_id) and written beside every PMC collection in
    profiles/, so that a bench line can tell whether a collected figure describes the kernels it
    times (a change to the inflater does not stale the match finder's traffic figure)."""
    everything = SOURCES + [h for h in HEADERS]
    parts = ["all:" + _hash_files(everything)]
    for g, files in GROUPS.items():
        parts.append(g + ":" + _hash_files(files + SHARED))
    return ";".join(parts)


def id_component(build_id, group):
    """The `group` hash of a build id string (None if absent)."""
    for part in (build_id or "").split(";"):
        k, _, v = part.partition(":")
        if k == group:
            return v
    return None


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + \
           [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-DFLATE_HIP_BUILD_ID=\"%s\"" % source_hash(),
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + srcs + ["-o", LIB_PATH, "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
