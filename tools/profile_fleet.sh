#!/bin/bash
# Dev tool: per-kernel time of the config-5 closed loop (rocprofv3 --kernel-trace --stats of tools/fleet_run.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_fleet; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python3 $R/tools/fleet_run.py > $O/run.log 2>&1
f=$(find /tmp/prof_f -name '*kernel_stats.csv' | head -1); cp $f $O/fleet_kernel_stats.csv; grep -v amdgpu.ids $O/run.log | tail -3; head -12 $O/fleet_kernel_stats.csv | cut -c1-200
