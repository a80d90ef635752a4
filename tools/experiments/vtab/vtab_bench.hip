// vtab_bench.hip -- EXPERIMENT (round 6): one stream's REAL slot trace (slot_trace.cpp)
// replayed through the register-resident hash table of vtab.h, alone and beside the match finder's default launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/experiments/vtab/vtab_bench.hip -o build/exp/vtab_bench -ldl
//   build/exp/vtab_bench build/exp/slot_trace.bin [moonbit-flate_amd/lib/libflate_hip.so]
// Reports cycles per batch of the gather pass and of the insert loops (s_memtime, per wavefront, median over the grid),
// checks every lookup against the trace's expected values (and the trace itself through an LDS table), and -- with the
// library -- the match finder's time with and without these wavefronts beside it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "flate_hip.h"
#include "vtab.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Args {
  const uint32_t *words;   // 64 per batch
  const uint32_t *first;   // first batch of every stream, nstreams + 1
  uint16_t *out;           // 64 per batch: lookup results (written by the blocks < nstreams in their first repetition)
  uint64_t *stats;         // per block: {gather cycles, scatter cycles, batches, hw id}
  uint32_t nstreams, reps;
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(96))) void vt_reg_kernel(Args A) {
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x % A.nstreams;
  const uint32_t b0 = A.first[sid], b1 = A.first[sid + 1];
  uint64_t cg = 0, cs = 0, nb = 0;
  for (uint32_t rep = 0; rep < A.reps; ++rep) {
    vt_init();
    const bool check = rep == 0 && blockIdx.x < A.nstreams;
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t w = A.words[(size_t)b * 64 + lane];
      const uint32_t h = w & 16383u;
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      const uint32_t dw = vt_gather(h >> 7, (h & 63u) << 2);
      const uint32_t old = (w & 0x4000u) ? ((h & 64u) ? (dw >> 16) : (dw & 0xffffu)) : 0u;
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      if (check) A.out[(size_t)b * 64 + lane] = (uint16_t)old;
      const bool ins = (w & 0x8000u) != 0;
      const uint32_t hw = (h >> 7) | ((h & 63u) << 8) | (w & 0xffff0000u);
      const uint64_t mlo = __ballot(ins && !(h & 64u)), mhi = __ballot(ins && (h & 64u));
      const uint64_t t2 = __builtin_amdgcn_s_memtime();
      vt_scatter_half(mlo, hw, 0x0000ffffu);
      vt_scatter_half(mhi, hw, 0xffff0000u);
      const uint64_t t3 = __builtin_amdgcn_s_memtime();
      cg += t1 - t0;
      cs += t3 - t2;
      ++nb;
      asm volatile("" :: "v"(old));
    }
  }
  if (lane == 0) {
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint64_t *s = A.stats + (size_t)blockIdx.x * 4;
    s[0] = cg; s[1] = cs; s[2] = nb; s[3] = ((uint64_t)(xcc & 15u) << 32) | hwid;
  }
}

// the same trace through an LDS table (lane-parallel gather and commit, as the LDS-table blocks do): validates the trace
__global__ __launch_bounds__(64) void vt_lds_kernel(Args A) {
  __shared__ uint16_t table[16384];
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x % A.nstreams;
  const uint32_t b0 = A.first[sid], b1 = A.first[sid + 1];
  uint64_t cg = 0, nb = 0;
  for (uint32_t rep = 0; rep < A.reps; ++rep) {
    for (int i = lane; i < 16384; i += 64) table[i] = 0;
    __syncthreads();
    const bool check = rep == 0 && blockIdx.x < A.nstreams;
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t w = A.words[(size_t)b * 64 + lane];
      const uint32_t h = w & 16383u;
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      const uint32_t old = (w & 0x4000u) ? table[h] : 0u;
      asm volatile("" ::: "memory");
      if (check) A.out[(size_t)b * 64 + lane] = (uint16_t)old;
      if (w & 0x8000u) table[h] = (uint16_t)(w >> 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      cg += t1 - t0;
      ++nb;
      asm volatile("" :: "v"(old));
    }
    __syncthreads();
  }
  if (lane == 0) {
    uint64_t *s = A.stats + (size_t)blockIdx.x * 4;
    s[0] = cg; s[1] = 0; s[2] = nb; s[3] = 0;
  }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  if (argc < 2) { fprintf(stderr, "usage: vtab_bench <trace> [libflate_hip.so]\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint32_t hdr[4];
  if (fread(hdr, 4, 4, f) != 4 || hdr[0] != 0x56544142u) { fprintf(stderr, "bad trace\n"); return 2; }
  const uint32_t nstreams = hdr[1], nbatch = hdr[2];
  std::vector<uint32_t> first(nstreams + 1), words((size_t)nbatch * 64);
  std::vector<uint16_t> expect((size_t)nbatch * 64), got((size_t)nbatch * 64);
  if (fread(first.data(), 4, first.size(), f) != first.size() || fread(words.data(), 4, words.size(), f) != words.size() ||
      fread(expect.data(), 2, expect.size(), f) != expect.size()) { fprintf(stderr, "short trace\n"); return 2; }
  fclose(f);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; trace: %u streams, %u batches\n", prop.gcnArchName, cus, nstreams, nbatch);

  uint32_t *d_words, *d_first; uint16_t *d_out; uint64_t *d_stats;
  const int max_blocks = cus * 8;
  CK(hipMalloc(&d_words, words.size() * 4)); CK(hipMalloc(&d_first, first.size() * 4));
  CK(hipMalloc(&d_out, got.size() * 2)); CK(hipMalloc(&d_stats, (size_t)max_blocks * 32));
  CK(hipMemcpy(d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
  Args A{d_words, d_first, d_out, d_stats, nstreams, 1};

  auto check = [&](const char *what) {
    CK(hipMemcpy(got.data(), d_out, got.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, firstbad = 0;
    for (size_t i = 0; i < got.size(); ++i) if (got[i] != expect[i]) { if (!bad) firstbad = i; ++bad; }
    printf("%s: %zu of %zu lookup results differ from the trace's expected values%s\n", what, bad, got.size(), bad ? " <-- WRONG" : " (all equal)");
    if (bad) printf("   first at batch %zu lane %zu: got %u expected %u word %08x\n", firstbad / 64, firstbad % 64, got[firstbad], expect[firstbad], words[firstbad]);
    return bad == 0;
  };
  auto stats = [&](int blocks, const char *what, bool placement) {
    std::vector<uint64_t> s((size_t)blocks * 4);
    CK(hipMemcpy(s.data(), d_stats, s.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> g, sc;
    for (int i = 0; i < blocks; ++i) if (s[i * 4 + 2]) { g.push_back((double)s[i * 4] / s[i * 4 + 2]); sc.push_back((double)s[i * 4 + 1] / s[i * 4 + 2]); }
    printf("%s: cycles per batch, median over %zu wavefronts: gather %.0f, inserts %.0f, table work %.0f (min %.0f max %.0f)\n", what, g.size(),
           med(g), med(sc), med(g) + med(sc), *std::min_element(g.begin(), g.end()) + *std::min_element(sc.begin(), sc.end()),
           *std::max_element(g.begin(), g.end()) + *std::max_element(sc.begin(), sc.end()));
    if (placement) {  // wavefronts per SIMD: HW_ID [5:4] simd, [11:8] cu, [12] sh, [15:13] se; XCC id beside it
      std::vector<uint32_t> key;
      for (int i = 0; i < blocks; ++i) { const uint32_t hw = (uint32_t)s[i * 4 + 3], xcc = (uint32_t)(s[i * 4 + 3] >> 32); key.push_back((xcc << 16) | (hw & 0xff30u)); }
      std::sort(key.begin(), key.end());
      int hist[9] = {0}; size_t i = 0; int simds = 0;
      while (i < key.size()) { size_t j = i; while (j < key.size() && key[j] == key[i]) ++j; hist[std::min<size_t>(j - i, 8)]++; ++simds; i = j; }
      printf("   placement (end of kernel): %d SIMDs hold these wavefronts; SIMDs with 1/2/3/4+ of them: %d/%d/%d/%d\n", simds, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7] + hist[8]);
    }
  };

  // 1. the trace through an LDS table
  memset(got.data(), 0xff, got.size() * 2);
  CK(hipMemset(d_out, 0xff, got.size() * 2));
  hipLaunchKernelGGL(vt_lds_kernel, dim3(cus * 4), dim3(64), 0, 0, A);
  CK(hipDeviceSynchronize());
  check("LDS table");
  stats(cus * 4, "LDS table, 4 wavefronts per CU", false);
  // 2. the register table, alone
  CK(hipMemset(d_out, 0xff, got.size() * 2));
  hipLaunchKernelGGL(vt_reg_kernel, dim3(cus * 4), dim3(64), 0, 0, A);
  CK(hipDeviceSynchronize());
  const bool ok = check("register table");
  stats(cus * 4, "register table, 4 wavefronts per CU, alone", true);
  for (int per_cu : {1, 2, 8}) {
    hipLaunchKernelGGL(vt_reg_kernel, dim3(cus * per_cu), dim3(64), 0, 0, A);
    CK(hipDeviceSynchronize());
    char name[96]; snprintf(name, sizeof name, "register table, %d wavefronts per CU, alone", per_cu);
    stats(cus * per_cu, name, true);
  }
  if (argc < 3 || !ok) return ok ? 0 : 1;

  // 3. beside the match finder's default launch (16384 x 65536 B of S-text, resident in HBM)
  void *lib = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
#define SYM(name) auto p_##name = (decltype(&name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing " #name "\n"); return 2; }
  SYM(flate_hip_init) SYM(flate_hip_destroy) SYM(flate_hip_deflate_fast_batch) SYM(flate_hip_set_profiling) SYM(flate_hip_last_timing)
  SYM(flate_hip_synth_fill) SYM(flate_hip_deflate_bound) SYM(flate_hip_set_option) SYM(flate_hip_last_resident_share)
  const uint32_t n = 16384; const uint64_t blen = 65536;
  std::vector<uint8_t> host((size_t)n * blen);
  p_flate_hip_synth_fill(FLATE_SYNTH_TEXT, 0x5EED0001ull, 0, n, blen, host.data(), 16);
  uint8_t *d_in, *d_cmp; const uint64_t cap = (uint64_t)n * blen * 3 / 4;
  CK(hipMalloc(&d_in, host.size())); CK(hipMalloc(&d_cmp, cap));
  CK(hipMemcpy(d_in, host.data(), host.size(), hipMemcpyHostToDevice));
  std::vector<uint64_t> in_off(n + 1), out_off(n + 1);
  for (uint32_t i = 0; i <= n; ++i) in_off[i] = i * blen;
  flate_hip_ctx *ctx = nullptr;
  if (p_flate_hip_init(0, &ctx)) { fprintf(stderr, "flate_hip_init failed\n"); return 2; }
  p_flate_hip_set_profiling(ctx, 1);
  auto run = [&](int calls, const char *what) {
    std::vector<double> lz, hp;
    for (int k = 0; k < calls; ++k) {
      const int rc = p_flate_hip_deflate_fast_batch(ctx, d_in, in_off.data(), n, d_cmp, cap, out_off.data(), FLATE_HIP_DEVICE_PTRS);
      if (rc) { fprintf(stderr, "deflate rc %d\n", rc); exit(2); }
      float ms[FLATE_HIP_STAGE_COUNT];
      p_flate_hip_last_timing(ctx, ms, FLATE_HIP_STAGE_COUNT);
      if (k) { lz.push_back(ms[FLATE_HIP_STAGE_LZ77]); hp.push_back(ms[FLATE_HIP_STAGE_HUFF_PACK]); }
    }
    uint32_t res = 0, q = 0;
    p_flate_hip_last_resident_share(ctx, &res, &q);
    printf("%s: match finder %.2f ms median (min %.2f max %.2f), entropy %.2f ms; LDS-table blocks took %u of %u streams; %llu compressed bytes\n", what, med(lz),
           *std::min_element(lz.begin(), lz.end()), *std::max_element(lz.begin(), lz.end()), med(hp), res, q, (unsigned long long)out_off[n]);
    return med(lz);
  };
  const double base = run(8, "match finder alone");
  if (argc > 3 && !strcmp(argv[3], "mf-only")) { p_flate_hip_destroy(ctx); return 0; }
  hipStream_t s2;
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int per_cu : {4, 2, 1}) {
    Args B = A;
    B.reps = 200;  // (about 1.5 ms per repetition: the wavefronts outlast the eight calls below)
    CK(hipEventRecord(e0, s2));
    hipLaunchKernelGGL(vt_reg_kernel, dim3(cus * per_cu), dim3(64), 0, s2, B);
    CK(hipEventRecord(e1, s2));
    char name[128]; snprintf(name, sizeof name, "match finder beside %d register-table wavefronts per CU", per_cu);
    const double with = run(8, name);
    const bool still = hipEventQuery(e1) == hipErrorNotReady;
    CK(hipStreamSynchronize(s2));
    float vt_ms = 0; CK(hipEventElapsedTime(&vt_ms, e0, e1));
    printf("   register-table kernel: %.1f ms for %u repetitions (%s when the last call returned); match finder %+.1f %%\n", vt_ms, B.reps,
           still ? "still running" : "ALREADY FINISHED: the overlap was partial", 100.0 * (with - base) / base);
    snprintf(name, sizeof name, "register table, %d wavefronts per CU, beside the match finder", per_cu);
    stats(cus * per_cu, name, true);
  }
  run(4, "match finder alone again");
  p_flate_hip_destroy(ctx);
  return 0;
}
