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


def source_hash(flags_by_source=None):
    """'all:<h>;lz77:<h>;huff:<h>;inflate:<h>[;flags:<h>]' -- hashes over the sources the library is built from:
    everything, and per kernel family (its own files + the shared headers and the ABI file).
    Compiled into the library (flate_hip_build_id) and written beside every PMC collection in
    profiles/, so that a bench line can tell whether a collected figure describes the kernels it
    times (a change to the inflater does not stale the match finder's traffic figure).  A build with
    per-source flags (the TEST build, experiment builds) carries a 'flags:' component, so that it never
    reports the product library's id."""
    import hashlib
    everything = SOURCES + [h for h in HEADERS]
    parts = ["all:" + _hash_files(everything)]
    for g, files in GROUPS.items():
        parts.append(g + ":" + _hash_files(files + SHARED))
    if flags_by_source:
        desc = ";".join("%s=%s" % (k, " ".join(v)) for k, v in sorted(flags_by_source.items()))
        parts.append("flags:" + hashlib.sha256(desc.encode()).hexdigest()[:8])
    return ";".join(parts)


def id_component(build_id, group):
    """The `group` hash of a build id string (None if absent)."""
    for part in (build_id or "").split(";"):
        k, _, v = part.partition(":")
        if k == group:
            return v
    return None


def _stale(lib_path=None):
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + \
           [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


OBJ_DIR = os.path.join(ROOT, "build", "obj")
# The TEST build: the same objects, except that gather.hip is compiled with -DFLATE_HIP_TEST_BUILD, which is
# the only thing that lets FLATE_HIP_TEST_TRANSPORT replace RCCL (tests/rehearsal_transport/).  The product
# library has no such hook; tests that need it load this file through FLATE_HIP_LIB.
TEST_LIB_PATH = os.path.join(LIB_DIR, "libflate_hip_testbuild.so")
TEST_BUILD_FLAGS = {"gather.hip": ["-DFLATE_HIP_TEST_BUILD"]}


def _compile_objects(hipcc, flags_by_source, verbose):
    """One object per source, compiled in parallel and cached under build/obj by a hash of the source,
    the headers and the flags (a kernel experiment recompiles one file, not nine)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_hash = _hash_files([h for h in HEADERS])
    base = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    build_id = {"flate_api.hip": ["-DFLATE_HIP_BUILD_ID=\"%s\"" % source_hash(flags_by_source)]}  # (the one file that reports it)
    jobs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        flags = base + build_id.get(src, []) + flags_by_source.get(src, [])
        key = hashlib.sha256((hdr_hash + "\0" + " ".join(flags) + "\0").encode() + open(path, "rb").read()).hexdigest()[:16]
        obj = os.path.join(OBJ_DIR, "%s-%s.o" % (os.path.splitext(src)[0], key))
        jobs.append((path, obj, flags))

    def one(job):
        path, obj, flags = job
        if not os.path.exists(obj):
            tmp = "%s.%d.tmp" % (obj, os.getpid())  # (two builders at once never write one file)
            cmd = [hipcc] + flags + ["-c", path, "-o", tmp]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        return list(pool.map(one, jobs))


def _link(hipcc, objs, out, verbose):
    tmp = "%s.%d.tmp" % (out, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp, "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _link(hipcc, _compile_objects(hipcc, {}, verbose), LIB_PATH, verbose)
    return LIB_PATH


def build_test(force=False, verbose=False):
    """libflate_hip_testbuild.so: see TEST_LIB_PATH.  Test infrastructure, built by the tests that need it
    (and by __graft_entry__.build(), so that it travels to the GPU box)."""
    if not force and not _stale(TEST_LIB_PATH):
        return TEST_LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _link(hipcc, _compile_objects(hipcc, TEST_BUILD_FLAGS, verbose), TEST_LIB_PATH, verbose)
    return TEST_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--test" in sys.argv:
        print(build_test(force="--force" in sys.argv, verbose=True))
