#!/bin/bash
# The host-pointer calls against the number of HIP hardware queues and the priorities of the guest / copy
# streams (kernels and copies that must run side by side: see flate_hip_init, host_pipe_streams).
for q in 4 8; do for p in normal high; do echo "GPU_MAX_HW_QUEUES=$q priorities=$p"
  GPU_MAX_HW_QUEUES=$q FLATE_HIP_GUEST_STREAM_PRIORITY=$p python3 tools/experiments/host_path.py deflate 2>&1 | grep "lanes=2 groups=8 \|lanes=1 groups=4 \|pageable"
done; done
