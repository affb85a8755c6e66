#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command      -> <out>/kernel_stats.csv, bench line, launch durations
#   2. separate --pmc passes (never combined with trace domains other than --kernel-trace)
#        FETCH_SIZE | WRITE_SIZE                                   HBM traffic (gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md)
#        SQ_INSTS_VALU SQ_ACTIVE_INST_VALU | SQ_WAVE_CYCLES SQ_BUSY_CYCLES      issue / occupancy
#        SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 SQ_INSTS_VALU_MFMA_MOPS_F64       executed FP64 flops
#      -> <out>/pmc.csv (mean per launch over the timed launches) and <out>/traffic.json (what bench.py reads from
#         profiles/traffic.json: HBM bytes and executed FP64 flops per launch of this workload)
# Raw traces are summarised and deleted (gpurun copies back at most 64 MiB).
# Usage: bash tools/profile_round.sh <tag> [extra bench.py args]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift || true
EXTRA="$@"
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-cpu-baseline --no-other-configs $EXTRA > $O/bench_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob('/tmp/prof_stats/**/*kernel_trace.csv', recursive=True)[0]
import os, re
RX = re.compile(os.environ.get('KERNEL_RX', r'plan_step_kernel<.*true>'))      # the timed kernel (bench.py also launches the kernel that keeps every row, once, for frac_no_presolve)
rows = [r for r in csv.DictReader(open(f)) if RX.search(r['Kernel_Name'])]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows][-51:-1]      # the timed launches (the very last launch of this kernel is bench.py's answer-collecting one, with diagnostics)
vg = rows[-1]
open(O + '/bench_launches.txt', 'w').write('last %d plan_step_kernel launches: mean %.1f us, min %.1f, max %.1f; %s VGPR %s accum %s SGPR %s LDS %s scratch %s\n' % (
    len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, vg['Kernel_Name'][:60], vg.get('VGPR_Count', '?'), vg.get('Accum_VGPR_Count', '?'),
    vg.get('SGPR_Count', '?'), vg.get('LDS_Block_Size', '?'), vg.get('Scratch_Size', '?')))
PY
rm -rf /tmp/prof_stats
echo "counter,mean_per_launch_over_the_10_timed_launches" > $O/pmc.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_pmc -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 $EXTRA > $O/bench_pmc.log 2>&1
  python3 - "$O" <<'PY'
import csv, glob, sys, collections, os, re
RX = re.compile(os.environ.get('KERNEL_RX', r'plan_step_kernel<.*true>'))
O = sys.argv[1]
f = glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if RX.search(r['Kernel_Name']):
        per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
with open(O + '/pmc.csv', 'a') as o:
    for name, disp in per.items():
        vals = [v for k, v in sorted(disp.items(), key=lambda kv: int(kv[0]))][-11:-1]
        o.write('%s,%.1f\n' % (name, sum(vals) / len(vals)))
PY
done
rm -rf /tmp/prof_pmc
python3 - "$O" <<'PY'
import csv, json, sys
O = sys.argv[1]
c = {r[0]: float(r[1]) for r in list(csv.reader(open(O + '/pmc.csv')))[1:]}
line = [l for l in open(O + '/bench_stats.log') if l.startswith('{')][-1]
b = json.loads(line)
key = 'N%d_obs%d_B%d' % (b['config']['horizon'], b['config']['obstacles'], b['config']['batch_rank0'])
flops = 64.0 * (2 * c.get('SQ_INSTS_VALU_FMA_F64', 0) + c.get('SQ_INSTS_VALU_MUL_F64', 0) + c.get('SQ_INSTS_VALU_ADD_F64', 0) + c.get('SQ_INSTS_VALU_TRANS_F64', 0))
rec = {key: {'hbm_bytes': int((2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024), 'executed_fp64_flops': flops,
             'fp64_wave_instructions': {k: c.get('SQ_INSTS_VALU_' + k + '_F64', 0) for k in ('FMA', 'MUL', 'ADD', 'TRANS')},
             'mfma_f64_mops': c.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0)},
       '_note': 'per launch, from separate rocprofv3 --pmc passes (tools/profile_round.sh): hbm_bytes = (FETCH_SIZE*2 + WRITE_SIZE) KB '
                '(gfx950 FETCH_SIZE x2 rule of MI355X_MICROARCH.md; 8-byte scalar accesses are outside the guide\'s calibration); '
                'executed_fp64_flops = 64 lanes x (2 FMA + MUL + ADD + TRANS) FP64 VALU wave-instructions (lanes of finished '
                'groups are masked off but counted: an upper bound of the useful flops)'}
json.dump(rec, open(O + '/traffic.json', 'w'), indent=1)
print(json.dumps(rec))
PY
cat $O/bench_launches.txt $O/pmc.csv
