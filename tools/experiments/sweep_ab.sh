#!/bin/bash
# EXPERIMENT (round 6): how often the multi-window kernels sweep their 16-bit modular table (kSweepEvery) against how far
# a sparse batch may reach (kSpanMax) -- the two share one budget (csrc/lz77_device.h).  Config 3 stage times per variant
# library (tools/build_variant.sh sw* -DFLATE_LZ_SWEEP_EVERY=.. -DFLATE_LZ_SPAN_MAX=.. [-DFLATE_LZ_MARKER_BACK=..]), two rounds.
for rep in 1 2; do
  for v in swbase swA swB swD; do
    FLATE_HIP_LIB=build/exp/lib$v.so python3 tools/experiments/c3_time.py 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
