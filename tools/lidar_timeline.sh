#!/bin/bash
# Dev tool: kernel timeline of the config-5 scan (ranking kernels + scan) from a rocprofv3 kernel trace: durations and the gaps
# between the three kernels of one call.  Run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-cfg5}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 $R/tools/run_workload.py $WL 10 > /tmp/prof_tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)) if 'lidar_' in r['Kernel_Name']))
calls = []
i = 0
while i + 2 < len(rows):
    if 'weight' in rows[i][2] and 'order' in rows[i + 1][2] and 'sense' in rows[i + 2][2]:
        calls.append(rows[i:i + 3]); i += 3
    else:
        i += 1
calls = calls[-10:]
m = lambda v: sum(v) / len(v) / 1e3
print('calls %d: weight %.1f us, gap %.1f, order %.1f, gap %.1f, scan %.1f; first start to last end %.1f us; call to call %.1f us' % (
    len(calls), m([c[0][1] - c[0][0] for c in calls]), m([c[1][0] - c[0][1] for c in calls]), m([c[1][1] - c[1][0] for c in calls]),
    m([c[2][0] - c[1][1] for c in calls]), m([c[2][1] - c[2][0] for c in calls]), m([c[2][1] - c[0][0] for c in calls]),
    m([b[0][0] - a[0][0] for a, b in zip(calls[:-1], calls[1:])])))
PY
