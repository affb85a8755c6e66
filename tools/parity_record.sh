#!/bin/bash
# Writes the observed GPU-vs-oracle agreement figures of every parity comparison of tests/test_gpu_configs.py
# (statuses, iterations, footsteps, decisive share, active-set mismatches, UNCERTIFIED counts, next to the bars the
# tests hold them to) into profiles/<tag>_parity.json.  Run on the GPU box from the repo root:
#   bash tools/parity_record.sh r04
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/${TAG}_parity.json
LIPMPC_PARITY_RECORD=$R/gpurun_out/${TAG}_parity.json python3 -m pytest $R/tests/test_gpu_configs.py -m gpu -q -x -k "config2 or config3 or config4 or bench_inputs or split_launch" > $R/gpurun_out/${TAG}_parity.log 2>&1
tail -3 $R/gpurun_out/${TAG}_parity.log
