// checksum.hip -- Adler-32 and CRC-32 of a batch of streams, for the container formats that wrap a raw
// DEFLATE stream ("optional gzip/zlib wrappers (absent from reference)"; RFC 1950 section 8.2,
// RFC 1952 section 8).  The reference has neither; the checker is oracle/checksum.c, pinned against zlib.
//
// Both checksums are linear in the right sense, so the work is cut into PIECES of 64 KiB regardless of how
// long a stream is (one stream of 1 GiB and 16384 streams of 64 KiB fill the chip alike):
//   checksum_piece_kernel  one wavefront per piece.  A FULL piece is read in 64 coalesced rows of 1 KiB, lane
//       L taking bytes [16 L, 16 L + 16) of every row:
//       CRC-32   the lane carries the raw (zero-initialised, hence linear) CRC of its sixteen-byte pieces as if
//                they were contiguous: per row a multiplication by x^(8 * 1008) (the gap; four table lookups)
//                and slicing-by-4 over the row's four dwords (sixteen lookups; eight 1 KiB tables in LDS); at
//                the end a multiplication by x^(8 * bytes behind the lane's last piece), XOR over the lanes,
//                and the pre-/post-inversion as a term that depends on the length only
//                (crc(A || B) = crc(A) * x^(8 |B|) + crc(B): zlib's crc32_combine);
//       Adler-32 sum of the bytes and sum of (index * byte) per lane (v_dot4), added over the lanes.
//       The last, shorter piece of a stream: lane L takes bytes [1024 L, 1024 L + 1024) of it instead;
//   checksum_fold_kernel   one wavefront per stream folds its pieces: 64 runs of consecutive pieces, one per
//       lane, then the lanes.
// One read of the input; what bounds the kernels is stated in DESIGN 4.7.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <new>
#include <stdexcept>
#include <vector>

#include <string>

#include "flate_hip.h"
#include "flate_kernels.h"

namespace flate {

namespace {

constexpr uint32_t kPiece = 65536;  // bytes per wavefront and step
constexpr uint32_t kChunk = 1024;   // bytes per lane of it
constexpr uint32_t kPoly = 0xedb88320u;
constexpr uint32_t kAdlerMod = 65521u;

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0); zlib's multmodp
__host__ __device__ inline uint32_t multmodp(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1u)) == 0) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
  }
  return p;
}

struct X2n {
  uint32_t t[32];  // x^(2^n) mod P
};
inline X2n make_x2n() {
  X2n r;
  uint32_t p = 1u << 30;  // x^1
  r.t[0] = p;
  for (int n = 1; n < 32; ++n) r.t[n] = p = multmodp(p, p);
  return r;
}

// x^(n * 2^k) mod P
__device__ inline uint32_t x2nmodp(const X2n &T, uint64_t n, unsigned k) {
  uint32_t p = 1u << 31;  // x^0
  while (n) {
    if (n & 1u) p = multmodp(T.t[k & 31u], p);
    n >>= 1;
    ++k;
  }
  return p;
}

struct PieceParams {
  const uint8_t *in;
  const uint64_t *piece_off;  // absolute offset of every piece in `in`
  const uint32_t *piece_len;  // 1 .. kPiece
  uint32_t n_pieces;
  uint32_t want_crc, want_adler;
  uint32_t *crc;      // per piece
  uint32_t *asum;     // per piece: sum of its bytes
  uint64_t *wsum;     // per piece: sum of (index inside the piece) * byte
  X2n x2n;
};

__device__ inline uint32_t ld32u(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// my bytes [1024 L, 1024 L + 1024) of a piece shorter than kPiece (a stream's last one)
__device__ inline void short_piece(const PieceParams &P, const uint32_t (*T)[256], uint32_t pc, int lane) {
  const uint32_t len = P.piece_len[pc];
  const uint8_t *src = P.in + P.piece_off[pc];
  const uint32_t start = kChunk * (uint32_t)lane;
  const uint32_t clen = len > start ? (len - start < kChunk ? len - start : kChunk) : 0u;
  const uint8_t *p = src + start;
  uint32_t c = 0xffffffffu;
  uint32_t a = 0;  // sum of my bytes (<= 1024 * 255)
  uint32_t w = 0;  // sum of (index inside my chunk) * byte (< 2^28)
  uint32_t i = 0;
  for (; i + 16 <= clen; i += 16) {
    uint4 q;
    __builtin_memcpy(&q, p + i, 16);
    const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (P.want_crc) {
        c ^= d[k];
        c = T[3][c & 255u] ^ T[2][(c >> 8) & 255u] ^ T[1][(c >> 16) & 255u] ^ T[0][c >> 24];
      }
      if (P.want_adler) {
        const uint32_t s4 = __builtin_amdgcn_udot4(d[k], 0x01010101u, 0u, false);
        a += s4;
        w = __builtin_amdgcn_udot4(d[k], 0x03020100u, w, false) + (i + 4u * k) * s4;
      }
    }
  }
  for (; i < clen; ++i) {
    const uint32_t b = p[i];
    if (P.want_crc) c = T[0][(c ^ b) & 255u] ^ (c >> 8);
    a += b;
    w += i * b;
  }
  if (P.want_crc) {
    c = clen ? c ^ 0xffffffffu : 0u;  // (the CRC of no bytes is 0)
    if (clen) c = multmodp(x2nmodp(P.x2n, len - (start + clen), 3), c);
    for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
    if (lane == 0) P.crc[pc] = c;
  }
  if (P.want_adler) {
    uint64_t ww = (uint64_t)w + (uint64_t)start * a;
    uint32_t aa = a;
    for (int d = 32; d >= 1; d >>= 1) {
      aa += (uint32_t)__shfl_xor((int)aa, d);
      ww += (uint64_t)__shfl_xor((long long)ww, d);
    }
    if (lane == 0) {
      P.asum[pc] = aa;
      P.wsum[pc] = ww;
    }
  }
}

__global__ __launch_bounds__(256) void checksum_piece_kernel(PieceParams P) {
  // T: slicing-by-4 tables (RFC 1952 section 8's table and three shifted copies); S: the same four byte
  // positions multiplied by x^(8 * 1008) -- the bytes between two of a lane's sixteen-byte pieces
  __shared__ uint32_t T[4][256], S[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  {
    uint32_t c = (uint32_t)tid;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? kPoly ^ (c >> 1) : c >> 1;
    T[0][tid] = c;
    __syncthreads();
    uint32_t v = c;
    for (int k = 1; k < 4; ++k) {
      v = (v >> 8) ^ T[0][v & 255u];
      T[k][tid] = v;
    }
    if (P.want_crc) {
      const uint32_t xgap = x2nmodp(P.x2n, kChunk - 16u, 3);
      for (int k = 0; k < 4; ++k) S[k][tid] = multmodp(xgap, (uint32_t)tid << (8 * k));
    }
    __syncthreads();
  }
  // x^(8 * bytes behind my last sixteen-byte piece of a full piece), and the inversions' term of a full piece
  const uint32_t xlane = P.want_crc ? x2nmodp(P.x2n, 16u * (uint32_t)(63 - lane), 3) : 0u;
  const uint32_t inv_full = P.want_crc ? multmodp(x2nmodp(P.x2n, kPiece, 3), 0xffffffffu) ^ 0xffffffffu : 0u;
  const uint32_t waves = gridDim.x * 4u;
  for (uint32_t pc = blockIdx.x * 4u + (uint32_t)wid; pc < P.n_pieces; pc += waves) {
    if (P.piece_len[pc] != kPiece) {
      short_piece(P, T, pc, lane);
      continue;
    }
    const uint8_t *src = P.in + P.piece_off[pc] + 16u * (uint32_t)lane;
    uint32_t c = 0;            // raw CRC of my pieces so far, as if contiguous
    uint32_t a = 0;            // sum of my bytes (<= 64 * 16 * 255)
    uint32_t w1 = 0, w2 = 0;   // sum over rows of row * (row's bytes); sum of (index inside the 16) * byte
#pragma unroll 4
    for (uint32_t r = 0; r < kPiece / kChunk; ++r) {
      uint4 q;
      __builtin_memcpy(&q, src + kChunk * r, 16);
      const uint32_t d[4] = {q.x, q.y, q.z, q.w};
      if (P.want_crc) {
        c = S[0][c & 255u] ^ S[1][(c >> 8) & 255u] ^ S[2][(c >> 16) & 255u] ^ S[3][c >> 24];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c ^= d[k];
          c = T[3][c & 255u] ^ T[2][(c >> 8) & 255u] ^ T[1][(c >> 16) & 255u] ^ T[0][c >> 24];
        }
      }
      if (P.want_adler) {
        uint32_t ar = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t s4 = __builtin_amdgcn_udot4(d[k], 0x01010101u, 0u, false);
          ar += s4;
          w2 = __builtin_amdgcn_udot4(d[k], 0x03020100u, w2, false) + 4u * k * s4;
        }
        a += ar;
        w1 += r * ar;
      }
    }
    if (P.want_crc) {
      c = multmodp(xlane, c);
      for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
      if (lane == 0) P.crc[pc] = c ^ inv_full;
    }
    if (P.want_adler) {
      uint64_t ww = (uint64_t)kChunk * w1 + (uint64_t)(16u * (uint32_t)lane) * a + w2;
      uint32_t aa = a;
      for (int d = 32; d >= 1; d >>= 1) {
        aa += (uint32_t)__shfl_xor((int)aa, d);
        ww += (uint64_t)__shfl_xor((long long)ww, d);
      }
      if (lane == 0) {
        P.asum[pc] = aa;
        P.wsum[pc] = ww;
      }
    }
  }
}

struct FoldParams {
  const uint64_t *in_off;      // n_streams + 1
  const uint32_t *piece_base;  // n_streams + 1: first piece of every stream
  const uint32_t *piece_len;
  const uint32_t *crc;
  const uint32_t *asum;
  const uint64_t *wsum;
  uint32_t *out;  // per stream
  uint32_t n_streams;
  uint32_t want_crc;  // else Adler-32
  uint32_t max_pieces;  // pieces of the longest stream
  X2n x2n;
};

// One wavefront per stream: lane L folds the run of pieces [p0 + L r, p0 + (L + 1) r), then the lanes are
// folded (every piece but a stream's last is kPiece long, so a run's place in the stream is known).
__global__ __launch_bounds__(256) void checksum_fold_kernel(FoldParams P) {
  // K[k][b] = (b << 8k) * x^(8 * kPiece) mod P: a full piece's step of the CRC fold as four lookups instead of
  // a 32-step multiplication (one long stream: 16384 pieces, 256 per lane)
  __shared__ uint32_t K[4][256];
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
  const bool tables = P.want_crc && P.max_pieces > 256u;  // (uniform: worth the 1024 multiplications per block)
  if (tables) {
    const uint32_t xp = x2nmodp(P.x2n, kPiece, 3);
    for (int k = 0; k < 4; ++k) K[k][threadIdx.x] = multmodp(xp, (uint32_t)threadIdx.x << (8 * k));
    __syncthreads();
  }
  if (s >= P.n_streams) return;
  const uint32_t p0 = P.piece_base[s], p1 = P.piece_base[s + 1];
  const uint32_t np = p1 - p0, run = (np + 63u) / 64u;
  const uint32_t k0 = p0 + run * (uint32_t)lane < p1 ? p0 + run * (uint32_t)lane : p1;
  const uint32_t k1 = k0 + run < p1 ? k0 + run : p1;
  const uint64_t n = P.in_off[s + 1] - P.in_off[s];
  if (P.want_crc) {
    const uint32_t xpiece = x2nmodp(P.x2n, kPiece, 3);
    uint32_t c = 0;
    uint64_t end = (uint64_t)(k0 - p0) * kPiece;  // bytes of the stream in front of my run, then behind its pieces
    for (uint32_t k = k0; k < k1; ++k) {
      const uint32_t len = P.piece_len[k];
      if (len == kPiece && tables)
        c = K[0][c & 255u] ^ K[1][(c >> 8) & 255u] ^ K[2][(c >> 16) & 255u] ^ K[3][c >> 24];
      else
        c = multmodp(len == kPiece ? xpiece : x2nmodp(P.x2n, len, 3), c);
      c ^= P.crc[k];
      end += len;
    }
    if (k1 > k0) c = multmodp(x2nmodp(P.x2n, n - end, 3), c);
    for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
    if (lane == 0) P.out[s] = c;
  } else {
    // s1 = 1 + sum b;  s2 = n + sum (n - i) b_i = n + n * sum b - sum i b_i   (all mod 65521, RFC 1950 8.2)
    uint64_t sa = 0, sib = 0, base = (uint64_t)(k0 - p0) * kPiece;
    for (uint32_t k = k0; k < k1; ++k) {
      const uint64_t a = P.asum[k];
      sa = (sa + a) % kAdlerMod;
      sib = (sib + (base % kAdlerMod) * (a % kAdlerMod) + P.wsum[k] % kAdlerMod) % kAdlerMod;
      base += P.piece_len[k];
    }
    for (int d = 32; d >= 1; d >>= 1) {
      sa += (uint64_t)__shfl_xor((long long)sa, d);
      sib += (uint64_t)__shfl_xor((long long)sib, d);
    }
    sa %= kAdlerMod;
    sib %= kAdlerMod;
    const uint64_t nm = n % kAdlerMod;
    const uint32_t s1 = (uint32_t)((1u + sa) % kAdlerMod);
    const uint32_t s2 = (uint32_t)((nm + nm * sa + (uint64_t)kAdlerMod * kAdlerMod - sib) % kAdlerMod);
    if (lane == 0) P.out[s] = (s2 << 16) | s1;
  }
}

}  // namespace

}  // namespace flate

using namespace flate;

extern "C" int flate_hip_checksum_batch(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                                        uint32_t kind, uint32_t *out, uint32_t flags) {
  if (!c || !in_off || (n && !out) || (kind != FLATE_HIP_CHECKSUM_ADLER32 && kind != FLATE_HIP_CHECKSUM_CRC32))
    return FLATE_HIP_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)
    if (in_off[i + 1] < in_off[i]) return FLATE_HIP_E_INVALID;
  if (n == 0) return FLATE_HIP_OK;
  const uint64_t total = in_off[n];
  if (total && !in) return FLATE_HIP_E_INVALID;
  ctx_set_error(c, "");
  auto hip_fail = [&](const char *what) -> int {
    ctx_set_error(c, std::string(what) + ": " + hipGetErrorString(hipGetLastError()));
    return FLATE_HIP_E_HIP;
  };
  if (hipSetDevice(ctx_device(c)) != hipSuccess) return hip_fail("hipSetDevice");
  hipStream_t st = ctx_stream(c);
  // pieces.  (The vectors may throw: nothing may cross the extern "C" boundary -- see the catch at the end.)
  try {
  std::vector<uint64_t> poff;
  std::vector<uint32_t> plen, pbase(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) {
    pbase[i] = (uint32_t)plen.size();
    for (uint64_t o = in_off[i]; o < in_off[i + 1]; o += kPiece) {
      poff.push_back(o);
      plen.push_back((uint32_t)(in_off[i + 1] - o < kPiece ? in_off[i + 1] - o : kPiece));
      if (plen.size() >= 0xfffffff0u) return FLATE_HIP_E_TOO_LARGE;
    }
  }
  pbase[n] = (uint32_t)plen.size();
  const uint32_t np = (uint32_t)plen.size();
  const bool dev = (flags & FLATE_HIP_DEVICE_PTRS) != 0;
  // one grow-only scratch of the ctx, carved into the call's arrays (round 4 did nine hipMalloc / hipFree
  // pairs per call: hipFree drains the whole device, i.e. every other context and the host pipelines' lanes)
  struct Dev { void *p = nullptr; } d_in, d_poff, d_plen, d_pbase, d_ioff, d_crc, d_asum, d_wsum, d_out;
  {
    size_t at = 0;
    auto carve = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_poff = carve((size_t)np * 8), o_wsum = carve((size_t)np * 8), o_ioff = carve(((size_t)n + 1) * 8),
                 o_plen = carve((size_t)np * 4 + 4), o_pbase = carve(((size_t)n + 1) * 4 + 4), o_crc = carve((size_t)np * 4),
                 o_asum = carve((size_t)np * 4), o_out = carve((size_t)n * 4 + 4);
    void *base = nullptr;
    int rc = ctx_scratch(c, 0, at, &base);
    if (rc == FLATE_HIP_OK && !dev) rc = ctx_scratch(c, 1, total + 16, &d_in.p);
    if (rc != FLATE_HIP_OK) return rc;
    uint8_t *b8 = (uint8_t *)base;
    d_poff.p = b8 + o_poff, d_wsum.p = b8 + o_wsum, d_ioff.p = b8 + o_ioff, d_plen.p = b8 + o_plen;
    d_pbase.p = b8 + o_pbase, d_crc.p = b8 + o_crc, d_asum.p = b8 + o_asum, d_out.p = b8 + o_out;
  }
  // the index arrays travel through the ctx's pinned staging and its copy kernel, not through DMA
  // commands that queue behind whatever bulk copy another thread has in flight (flate_api.hip: ctl_up)
  {
    int rc = ctx_ctl_begin(c, (size_t)np * 12 + ((size_t)n + 1) * 12 + 4 * 512, (size_t)n * 4 + 512);
    if (rc == FLATE_HIP_OK) rc = ctx_ctl_up(c, d_poff.p, poff.data(), (size_t)np * 8);
    if (rc == FLATE_HIP_OK) rc = ctx_ctl_up(c, d_plen.p, plen.data(), (size_t)np * 4);
    if (rc == FLATE_HIP_OK) rc = ctx_ctl_up(c, d_pbase.p, pbase.data(), ((size_t)n + 1) * 4);
    if (rc == FLATE_HIP_OK) rc = ctx_ctl_up(c, d_ioff.p, in_off, ((size_t)n + 1) * 8);
    if (rc != FLATE_HIP_OK) return rc;
    if (!dev && total && hipMemcpyAsync(d_in.p, in, total, hipMemcpyHostToDevice, st) != hipSuccess)
      return hip_fail("hipMemcpyAsync (checksum input)");
  }
  const X2n x2n = make_x2n();
  if (np) {
    PieceParams P{};
    P.in = dev ? in : (const uint8_t *)d_in.p;
    P.piece_off = (const uint64_t *)d_poff.p;
    P.piece_len = (const uint32_t *)d_plen.p;
    P.n_pieces = np;
    P.want_crc = kind == FLATE_HIP_CHECKSUM_CRC32;
    P.want_adler = kind == FLATE_HIP_CHECKSUM_ADLER32;
    P.crc = (uint32_t *)d_crc.p;
    P.asum = (uint32_t *)d_asum.p;
    P.wsum = (uint64_t *)d_wsum.p;
    P.x2n = x2n;
    uint32_t blocks = (np + 3) / 4;
    const uint32_t cap = 8u * (uint32_t)ctx_num_cus(c);  // 32 wavefronts per CU
    if (blocks > cap) blocks = cap;
    ctx_stage_begin(c, FLATE_HIP_STAGE_CHECKSUM);
    hipLaunchKernelGGL(checksum_piece_kernel, dim3(blocks), dim3(256), 0, st, P);
  } else {
    ctx_stage_begin(c, FLATE_HIP_STAGE_CHECKSUM);
  }
  FoldParams F{};
  F.in_off = (const uint64_t *)d_ioff.p;
  F.piece_base = (const uint32_t *)d_pbase.p;
  F.piece_len = (const uint32_t *)d_plen.p;
  F.crc = (const uint32_t *)d_crc.p;
  F.asum = (const uint32_t *)d_asum.p;
  F.wsum = (const uint64_t *)d_wsum.p;
  F.out = (uint32_t *)d_out.p;
  F.n_streams = n;
  F.want_crc = kind == FLATE_HIP_CHECKSUM_CRC32;
  F.max_pieces = 0;
  for (uint32_t i = 0; i < n; ++i) F.max_pieces = pbase[i + 1] - pbase[i] > F.max_pieces ? pbase[i + 1] - pbase[i] : F.max_pieces;
  F.x2n = x2n;
  hipLaunchKernelGGL(checksum_fold_kernel, dim3((n + 3) / 4), dim3(256), 0, st, F);
  ctx_stage_end(c, FLATE_HIP_STAGE_CHECKSUM);
  if (hipGetLastError() != hipSuccess) return hip_fail("checksum kernels");
  {
    const int rc = ctx_ctl_down(c, out, d_out.p, (size_t)n * 4);
    if (rc != FLATE_HIP_OK) return rc;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return hip_fail("checksum read-back");
  ctx_ctl_finish(c);
  return ctx_stage_collect(c, FLATE_HIP_STAGE_CHECKSUM);
  } catch (const std::bad_alloc &) {
    ctx_set_error(c, "out of host memory (checksum piece index)");
    return FLATE_HIP_E_HIP;
  } catch (const std::exception &e) {
    ctx_set_error(c, e.what());
    return FLATE_HIP_E_INTERNAL;
  }
}
