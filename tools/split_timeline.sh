#!/bin/bash
# Dev tool: kernel timeline of one split launch of config 4 (rocprofv3 --kernel-trace): start offset and duration of the
# classification, binning and per-body kernels of the LAST workload launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 $R/tools/run_workload.py ${1:-cfg4} 4 > /tmp/prof_tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('classify', 'split_bin', 'solve_list', 'plan_step'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last launch group = from the last classify kernel on
last = max(i for i, r in enumerate(rows) if 'classify' in r['Kernel_Name']) if any('classify' in r['Kernel_Name'] for r in rows) else len(rows) - 1
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-60s start +%8.1f us  duration %8.1f us  end +%8.1f us  grid %s  scratch %s  vgpr %s accum %s' % (
        r['Kernel_Name'][:60], (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get('Grid_Size'), r.get('Scratch_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count')))
PY
tail -2 /tmp/prof_tl.log
