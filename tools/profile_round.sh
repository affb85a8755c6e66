#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command      -> <out>/kernel_stats.csv, bench line
#   2. separate --pmc passes (HBM traffic, VALU / wave cycles)   -> <out>/pmc.csv (mean per launch, last 10 launches)
# Raw traces are summarised and deleted (gpurun copies back at most 64 MiB).
# Usage: bash tools/profile_round.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob('/tmp/prof_stats/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'plan_step_kernel' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows][-50:]
open(O + '/bench_launches.txt', 'w').write('last %d plan_step_kernel launches: mean %.1f us, min %.1f, max %.1f\n' % (len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
PY
rm -rf /tmp/prof_stats
echo "counter,mean_per_launch_over_the_10_timed_launches" > $O/pmc.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_pmc -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_pmc.log 2>&1
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'plan_step_kernel' in r['Kernel_Name']:
        per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
with open(O + '/pmc.csv', 'a') as o:
    for name, disp in per.items():
        vals = [v for k, v in sorted(disp.items(), key=lambda kv: int(kv[0]))][-10:]
        o.write('%s,%.1f\n' % (name, sum(vals) / len(vals)))
PY
done
rm -rf /tmp/prof_pmc
cat $O/bench_launches.txt $O/pmc.csv
