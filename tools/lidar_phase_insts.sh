#!/bin/bash
# Dev tool: VALU / SALU / LDS wave-instructions of lidar_sense_kernel up to each phase stop (the -DLIPMPC_LIDAR_PHASES variant in
# variants/phases.so), config-5 bench batch: where the scan's instructions go.  Run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
test -f $R/variants/phases.so
export LIPMPC_LIB=$R/variants/phases.so
cd /tmp && export TMPDIR=/tmp
for s in 6 7 1 2 3 0; do
  export LIPMPC_LIDAR_STOP=$s
  rm -rf /tmp/prof_ph
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/prof_ph -- python3 $R/tools/run_workload.py cfg5 3 > /tmp/prof_ph.log 2>&1 || true
  python3 - $s <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/prof_ph/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'lidar_sense_kernel' in r['Kernel_Name']:
        per[r['Counter_Name']].append(float(r['Counter_Value']))
print('stop', sys.argv[1], {k: round(sum(v[-3:]) / 3 / 4096) for k, v in per.items()}, 'per robot')
PY
done
