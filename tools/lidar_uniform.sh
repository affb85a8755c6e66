#!/bin/bash
# Dev tool: tools/lidar_uniform.py per phase with the -DLIPMPC_LIDAR_PHASES variant loaded from its own path (LIPMPC_LIB)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
test -f $R/variants/phases.so
for s in ${LIDAR_STOPS:-1 2 3 0}; do LIPMPC_LIB=$R/variants/phases.so LIPMPC_LIDAR_STOP=$s python3 $R/tools/lidar_uniform.py 2>&1 | grep -v amdgpu.ids; done
