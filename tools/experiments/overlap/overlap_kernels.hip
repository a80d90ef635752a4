// overlap_kernels.hip -- EXPERIMENT (round 6): can anything run BESIDE the match finder's
// persistent launch (4 LDS-table blocks + 6 guests per CU = 128 of 128 LDS granules on every CU)?  A copy kernel
// shaped like an RCCL collective's (few workgroups, 256-512 threads, some LDS each, device-to-device) and a stamp
// kernel, launched from tools/experiments/overlap/overlap_bench.py through ctypes.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/experiments/overlap/overlap_kernels.hip -o build/exp/liboverlap.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void ov_stamp_kernel(unsigned long long *slot) { *slot = __builtin_amdgcn_s_memrealtime(); }

// stamps[0] = earliest block start, stamps[1] = latest block end (100 MHz constant clock), stamps[2] = blocks run
// nap: s_sleep units (64 cycles each) per loop iteration -- a collective's kernel is bound by its links, not by HBM: it
// holds its workgroup slots and LDS for the transfer's time while mostly waiting
__global__ void ov_copy_kernel(uint4 *dst, const uint4 *src, size_t n16, unsigned long long *stamps, int nap) {
  extern __shared__ uint32_t lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    atomicMin(stamps, t0);
    lds[0] = (uint32_t)t0;  // (the allocation is real: touched)
  }
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    for (int k = 0; k < nap; k += 100) __builtin_amdgcn_s_sleep(100);
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() + (lds[0] & 0u));
    atomicAdd(stamps + 2, 1ull);
  }
}

extern "C" int ov_stamp(void *stream, unsigned long long *slot) {
  hipLaunchKernelGGL(ov_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
  return (int)hipGetLastError();
}
extern "C" int ov_copy(void *stream, void *dst, const void *src, size_t bytes, int wgs, int threads, int lds_bytes,
                       unsigned long long *stamps, int nap) {
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute((const void *)ov_copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ov_copy_kernel, dim3(wgs), dim3(threads), lds_bytes < 4 ? 4 : lds_bytes, (hipStream_t)stream, (uint4 *)dst,
                     (const uint4 *)src, bytes / 16, stamps, nap);
  return (int)hipGetLastError();
}
