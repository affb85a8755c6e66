#!/bin/bash
# Round-4 evidence in one go (GPU box, repo root): headline bench + the kernels real callers run.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/profile_round.sh r04_final > $R/gpurun_out/prof_r04_final.log 2>&1
bash $R/tools/profile_workload.sh r04_cfg4 cfg4 "classify_kernel|split_bin_kernel|solve_list_kernel" > $R/gpurun_out/prof_r04_cfg4.log 2>&1
bash $R/tools/profile_workload.sh r04_rollout rollout "rollout_kernel" 5 > $R/gpurun_out/prof_r04_rollout.log 2>&1
bash $R/tools/profile_workload.sh r04_cfg5_scan cfg5 "lidar_" > $R/gpurun_out/prof_r04_cfg5_scan.log 2>&1
bash $R/tools/profile_workload.sh r04_cfg5_solve cfg5 "plan_step_kernel" > $R/gpurun_out/prof_r04_cfg5_solve.log 2>&1
tail -n 4 $R/gpurun_out/prof_r04_final.log $R/gpurun_out/prof_r04_cfg4.log $R/gpurun_out/prof_r04_rollout.log $R/gpurun_out/prof_r04_cfg5_scan.log $R/gpurun_out/prof_r04_cfg5_solve.log
