#!/bin/bash
# Dev tool: phase split of the LiDAR kernel on the GPU box with the -DLIPMPC_LIDAR_PHASES variant (variants/phases.so,
# built by hand: see tools/README.md), loaded from its own path (LIPMPC_LIB): the shipped library is never touched.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
test -f $R/variants/phases.so
for B in 256 4096; do for s in 1 2 4 5 3 0; do LIPMPC_LIB=$R/variants/phases.so LIPMPC_LIDAR_STOP=$s python3 $R/tools/lidar_phases.py $B 2>&1 | grep -v amdgpu.ids; done; done
