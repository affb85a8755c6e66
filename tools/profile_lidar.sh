#!/bin/bash
# PMC counters of lidar_sense_kernel (tools/lidar_phases.py workload: 4096 robots of the config-5 bench map), one pass per
# counter group.  Usage: bash tools/profile_lidar.sh [B]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-4096}
O=$R/gpurun_out/prof_lidar
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "counter,mean_per_launch_B$B" > $O/lidar_pmc.csv
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_ANY"; do
  rm -rf /tmp/prof_l
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_l -- python3 $R/tools/lidar_phases.py $B > $O/run.log 2>&1 || { echo "pass $c failed"; tail -3 $O/run.log; continue; }
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
fs = glob.glob('/tmp/prof_l/**/*counter_collection.csv', recursive=True)
if fs:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if 'lidar_sense_kernel' in r['Kernel_Name']:
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    with open(O + '/lidar_pmc.csv', 'a') as o:
        for name, disp in per.items():
            vals = list(disp.values())[-10:]
            o.write('%s,%.1f\n' % (name, sum(vals) / len(vals)))
PY
done
rm -rf /tmp/prof_l
cat $O/lidar_pmc.csv
