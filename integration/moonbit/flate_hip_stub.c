/*
 * flate_hip_stub.c -- the few lines of C the MoonBit native backend needs next to
 * libflate_hip.so.  MoonBit's `extern "C"` cannot take
 * a pointer-to-pointer and has no null test for an #external type, so the four constructors
 * (ctx, comm, stream, inflate stream) and their null tests are wrapped, the size-only inflate pass fixes its
 * NULL arguments and the piecewise read packs its three results into one array; everything else binds
 * include/flate_hip.h directly.  UNVERIFIED with moon (not available in the build image); this file
 * itself is compiled by tests/test_library_abi.py to keep it in step with the header.
 */
#include "flate_hip.h"

/* -> flate_hip_ctx* or NULL; *rc_out (optional) receives the error code */
flate_hip_ctx *flate_hip_mbt_ctx_new(int device) {
  flate_hip_ctx *c = 0;
  if (flate_hip_init(device, &c) != FLATE_HIP_OK) return 0;
  return c;
}

/* MoonBit has no null test for an #external type */
int flate_hip_mbt_ctx_is_null(const flate_hip_ctx *c) { return c == 0; }

/* (FixedArray[Byte] / FixedArray[UInt64] / FixedArray[Int] arrive as plain pointers, so
 * flate_hip_deflate_fast_batch, flate_hip_inflate_batch, flate_hip_deflate_fast_spliced,
 * flate_hip_stream_write and the gather calls are bound by the .mbt file directly: no wrapper.) */

/* -- exchange step (multi-GPU): the communicator constructor returns its pointer instead of
 * writing it through a pointer-to-pointer; the gather calls bind include/flate_hip.h directly. -- */
flate_hip_comm *flate_hip_mbt_comm_new(flate_hip_ctx *c, const uint8_t *unique_id, int rank, int world) {
  flate_hip_comm *cm = 0;
  if (flate_hip_comm_init(c, unique_id, rank, world, &cm) != FLATE_HIP_OK) return 0;
  return cm;
}

int flate_hip_mbt_comm_is_null(const flate_hip_comm *cm) { return cm == 0; }

/* -- one long stream written in pieces; the size-only inflate pass -- */
flate_hip_stream *flate_hip_mbt_stream_new(flate_hip_ctx *c, uint32_t flags) {
  flate_hip_stream *st = 0;
  if (flate_hip_stream_open(c, flags, &st) != FLATE_HIP_OK) return 0;
  return st;
}

int flate_hip_mbt_stream_is_null(const flate_hip_stream *st) { return st == 0; }

int flate_hip_mbt_inflate_sizes(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                                uint64_t *out_len, int32_t *status, int64_t *err_off) {
  return flate_hip_inflate_batch(c, in, in_off, n, 0, 0, out_len, status, err_off, FLATE_HIP_SIZE_ONLY);
}

/* -- one long stream decoded in pieces: constructor + null test, and the three results of a read
 * packed into one array (MoonBit passes no pointers to scalars) -- */
flate_hip_inflate_stream *flate_hip_mbt_inflate_stream_new(flate_hip_ctx *c) {
  flate_hip_inflate_stream *st = 0;
  if (flate_hip_inflate_stream_open(c, &st) != FLATE_HIP_OK) return 0;
  return st;
}

int flate_hip_mbt_inflate_stream_is_null(const flate_hip_inflate_stream *st) { return st == 0; }

/* res[0] = bytes of `in` used, res[1] = bytes written to out, res[2] = err_off (int64 bits) */
int flate_hip_mbt_inflate_stream_read(flate_hip_inflate_stream *st, const uint8_t *in, uint64_t in_len, int final_in,
                                      uint8_t *out, uint64_t out_cap, uint64_t *res) {
  int64_t eoff = -1;
  const int rc = flate_hip_inflate_stream_read(st, in, in_len, final_in, out, out_cap, &res[0], &res[1], &eoff);
  res[2] = (uint64_t)eoff;
  return rc;
}
