#!/bin/bash
# rocprofv3 evidence for ONE workload of tools/run_workload.py (run on the GPU box from the repo root):
#   bash tools/profile_workload.sh <tag> <workload> "<kernel name regex>" [launches]
# 1. --kernel-trace --stats                     -> <out>/kernel_stats.csv, launches.txt (per-kernel mean duration of the timed launches)
# 2. separate --pmc passes (never combined with other trace domains), summed over the kernels matching the regex, per launch
#    of the workload                             -> <out>/pmc.csv, traffic.json (HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KB,
#    executed FP64 flops = 64 x (2 FMA + MUL + ADD + TRANS), wave-alive fraction = SQ_WAVE_CYCLES x 4 / (duration x clock x waves))
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; RX=$3; REPS=${4:-10}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/tools/run_workload.py $WL $REPS > $O/run_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python3 - "$O" "$RX" "$REPS" <<'PY'
import csv, glob, re, sys, collections
O, RX, REPS = sys.argv[1], re.compile(sys.argv[2]), int(sys.argv[3])
f = glob.glob('/tmp/prof_stats/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if RX.search(r['Kernel_Name'])]
per = collections.defaultdict(list)
for r in rows:
    per[r['Kernel_Name'][:90]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r))
with open(O + '/launches.txt', 'w') as o:
    for name, v in per.items():
        v = v[-REPS:]
        d = [e - s for s, e, _ in v]
        r = v[-1][2]
        o.write('%s: last %d launches mean %.1f us, min %.1f, max %.1f; VGPR %s accum %s SGPR %s LDS %s scratch %s grid %s\n' % (
            name, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, r.get('VGPR_Count', '?'), r.get('Accum_VGPR_Count', '?'),
            r.get('SGPR_Count', '?'), r.get('LDS_Block_Size', '?'), r.get('Scratch_Size', '?'), r.get('Grid_Size', '?')))
    # span of one workload launch: from the first matching kernel's start to the last one's end, over the last REPS groups
    allk = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
    n_per = max(1, len(per))
PY
rm -rf /tmp/prof_stats
echo "counter,mean_per_workload_launch" > $O/pmc.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_pmc -- python3 $R/tools/run_workload.py $WL $REPS > $O/run_pmc.log 2>&1
  python3 - "$O" "$RX" "$REPS" <<'PY'
import csv, glob, re, sys, collections
O, RX, REPS = sys.argv[1], re.compile(sys.argv[2]), int(sys.argv[3])
f = glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
kern = {}
for r in csv.DictReader(open(f)):
    if RX.search(r['Kernel_Name']):
        per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        kern[r['Dispatch_Id']] = r['Kernel_Name'][:60]
names = sorted(set(kern.values()))
with open(O + '/pmc.csv', 'a') as o:
    for name, disp in per.items():
        # per workload launch: the dispatches of the last REPS launches, all matching kernels summed
        by_k = collections.defaultdict(list)
        for k, v in sorted(disp.items(), key=lambda kv: int(kv[0])):
            by_k[kern[k]].append(v)
        tot = sum(sum(v[-REPS * max(1, len(v) // (REPS + 2)):]) / REPS for v in by_k.values()) if False else sum(sum(v[-REPS:]) / REPS for v in by_k.values())
        o.write('%s,%.1f\n' % (name, tot))
        for kn, v in by_k.items():
            o.write('%s[%s],%.1f\n' % (name, kn, sum(v[-REPS:]) / REPS))
PY
done
rm -rf /tmp/prof_pmc
python3 - "$O" <<'PY'
import csv, json, re, sys
O = sys.argv[1]
c = {r[0]: float(r[1]) for r in list(csv.reader(open(O + '/pmc.csv')))[1:] if '[' not in r[0]}
line = [l for l in open(O + '/run_stats.log') if l.startswith('{')][-1]
b = json.loads(line)
flops = 64.0 * (2 * c.get('SQ_INSTS_VALU_FMA_F64', 0) + c.get('SQ_INSTS_VALU_MUL_F64', 0) + c.get('SQ_INSTS_VALU_ADD_F64', 0) + c.get('SQ_INSTS_VALU_TRANS_F64', 0))
rec = {b['key']: {'hbm_bytes': int((2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024), 'write_bytes': int(c.get('WRITE_SIZE', 0) * 1024),
                  'executed_fp64_flops': flops, 'valu_wave_instructions': c.get('SQ_INSTS_VALU', 0), 'salu_wave_instructions': c.get('SQ_INSTS_SALU', 0),
                  'sq_wave_cycles': c.get('SQ_WAVE_CYCLES', 0), 'sq_waves': c.get('SQ_WAVES', 0), 'sq_wait_any': c.get('SQ_WAIT_ANY', 0),
                  'wait_any_over_wave_cycles': (c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']) if c.get('SQ_WAVE_CYCLES') else None,
                  'mfma_f64_mops': c.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0), 'workload': b}}
json.dump(rec, open(O + '/traffic.json', 'w'), indent=1)
print(json.dumps(rec))
PY
cat $O/launches.txt $O/pmc.csv
