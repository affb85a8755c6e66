#!/bin/bash
# Dev tool: duration of lidar_sense_kernel in a rocprofv3 kernel trace when it stops after each phase (the -DLIPMPC_LIDAR_PHASES
# variant in variants/phases.so; config-5 bench batch, ranked by the call): where the scan's TIME goes.  GPU box, repo root.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
test -f $R/variants/phases.so
export LIPMPC_LIB=$R/variants/phases.so
cd /tmp && export TMPDIR=/tmp
for s in 6 7 1 4 3 0; do
  export LIPMPC_LIDAR_STOP=$s
  rm -rf /tmp/prof_pt
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pt -- python3 $R/tools/run_workload.py cfg5 5 > /tmp/prof_pt.log 2>&1 || true
  python3 - $s <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_pt/**/*kernel_trace.csv', recursive=True)[0]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'lidar_sense_kernel' in r['Kernel_Name']][-5:]
print('stop %s: lidar_sense_kernel %.1f us' % (sys.argv[1], sum(d) / len(d) / 1e3))
PY
done
