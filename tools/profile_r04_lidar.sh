#!/bin/bash
# Round-4 evidence for the LiDAR front end at its final kernels (GPU box, repo root): rocprofv3 stats + PMC of the config-5 scan,
# the un-profiled bench lines, and the dev-tool records quoted in DESIGN.md.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
G=$R/gpurun_out
bash $R/tools/profile_workload.sh r04_cfg5_scan cfg5 "lidar_" > $G/prof_r04_cfg5_scan.log 2>&1
bash $R/tools/profile_workload.sh r04_cfg5_solve cfg5 "plan_step_kernel" > $G/prof_r04_cfg5_solve.log 2>&1
python3 $R/bench.py > $G/r04_final_bench.json 2> $G/r04_final_bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $G/r04_final_bench_steps20.json 2> $G/r04_final_bench_steps20.err
python3 $R/bench.py --no-cpu-baseline --all-configs > $G/r04_all_configs_bench.json 2> $G/r04_all_configs_bench.err
mkdir -p $G/r04_dev_tools
python3 $R/tools/lidar_order.py 2>&1 | grep -v amdgpu.ids > $G/r04_dev_tools/r04_lidar_order.txt
bash $R/tools/lidar_timeline.sh cfg5 2>&1 | tail -1 > $G/r04_dev_tools/r04_lidar_timeline.txt
bash $R/tools/lidar_timeline.sh cfg5maps 2>&1 | tail -1 >> $G/r04_dev_tools/r04_lidar_timeline.txt
if [ -f $R/variants/phases.so ]; then
  LIPMPC_LIB=$R/variants/phases.so LIPMPC_LIDAR_STOP=10 python3 $R/tools/lidar_wave_times.py 2>&1 | grep -v amdgpu.ids > $G/r04_dev_tools/r04_lidar_wave_times.txt
  LIPMPC_LIB=$R/variants/phases.so LIPMPC_LIDAR_STOP=9 python3 $R/tools/lidar_wave_times.py 2>&1 | grep -v amdgpu.ids >> $G/r04_dev_tools/r04_lidar_wave_times.txt
  LIPMPC_LIB=$R/variants/phases.so LIPMPC_LIDAR_STOP=8 python3 $R/tools/lidar_placement.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $G/r04_dev_tools/r04_lidar_placement.txt
  bash $R/tools/lidar_phase_insts.sh > $G/r04_dev_tools/r04_lidar_phase_insts.txt 2>&1
  LIDAR_STOPS="1 2 3 0" bash $R/tools/lidar_uniform.sh > $G/r04_dev_tools/r04_lidar_uniform.txt 2>&1
fi
tail -n 3 $G/prof_r04_cfg5_scan.log; cat $G/r04_final_bench.json | cut -c1-300
