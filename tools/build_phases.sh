#!/bin/bash
# Dev tool: variants/phases.so = the shipped objects with the LiDAR kernel rebuilt under -DLIPMPC_LIDAR_PHASES (tools/lidar_phases.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/humanoid-navigation-using-mpc-ldcbf_amd/csrc
mkdir -p $R/variants $C/build_phases
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLIPMPC_LIDAR_PHASES $EXTRA -c $C/lipmpc_lidar.hip -o $C/build_phases/lidar.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/phases.so $(ls $C/build/*.o | grep -v /lidar.o) $C/build_phases/lidar.o
