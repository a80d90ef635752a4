// rccl_selfsend.cpp -- EXPERIMENT (round 6): what does an RCCL kernel ask of a CU?  A one-rank communicator cannot run
// an all-gather kernel (RCCL copies), but a grouped ncclSend / ncclRecv to itself launches RCCL's real device kernel:
// under `rocprofv3 --kernel-trace` its row gives Grid_Size, Workgroup_Size, LDS_Block_Size, VGPR / SGPR counts.
//   hipcc -O2 tools/experiments/overlap/rccl_selfsend.cpp -o build/exp/rccl_selfsend -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "%s:%d error %d\n", __FILE__, __LINE__, (int)e_); exit(2); } } while (0)
int main() {
  CK(hipSetDevice(0));
  ncclUniqueId id;
  CK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  CK(ncclCommInitRank(&comm, 1, id, 0));
  const size_t bytes = 256u << 20;
  char *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 7, bytes));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int it = 0; it < 3; ++it) {
    CK(ncclGroupStart());
    CK(ncclSend(a, bytes, ncclChar, 0, comm, s));
    CK(ncclRecv(b, bytes, ncclChar, 0, comm, s));
    CK(ncclGroupEnd());
    CK(hipStreamSynchronize(s));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  CK(ncclGroupStart());
  CK(ncclSend(a, bytes, ncclChar, 0, comm, s));
  CK(ncclRecv(b, bytes, ncclChar, 0, comm, s));
  CK(ncclGroupEnd());
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  char h = 0; CK(hipMemcpy(&h, b + bytes - 1, 1, hipMemcpyDeviceToHost));
  printf("self send/recv of %zu MiB: %.3f ms (%.0f GB/s), last byte %d\n", bytes >> 20, ms, bytes / ms / 1e6, (int)h);
  ncclCommDestroy(comm);
  return 0;
}
