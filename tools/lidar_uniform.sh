#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/humanoid-navigation-using-mpc-ldcbf_amd/liblipmpc.so
cp $L /tmp/liblipmpc.keep && cp $R/variants/phases.so $L
for s in 1 2 3 0; do LIPMPC_LIDAR_STOP=$s python3 $R/tools/lidar_uniform.py 2>&1 | grep -v amdgpu.ids; done
cp /tmp/liblipmpc.keep $L
