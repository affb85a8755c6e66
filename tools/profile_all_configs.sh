#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --all-configs` (config 2 + the extras: config 4, config 5 scan/step/closed
# loop, rollout): per-kernel totals for profiles/.  Usage: bash tools/profile_all_configs.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_all
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_all -- python3 $R/bench.py --no-cpu-baseline --all-configs > $O/bench_all.log 2>&1
find /tmp/prof_all -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/all_configs_kernel_stats.csv
rm -rf /tmp/prof_all
head -12 $O/all_configs_kernel_stats.csv | cut -c1-160
