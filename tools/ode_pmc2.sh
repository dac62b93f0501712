#!/bin/bash
# SQ / SQC counters of one ionic kernel at 256^3 (tools/bench_kernels.py --only NAME), three passes: issue and wait
# cycles per unit, instruction mix, instruction-cache behaviour.  usage: bash tools/ode_pmc2.sh [torord|tp06] [library.so]
R=$PWD
WHAT=${1:-torord}
[ -n "$2" ] && export BEAT_HIP_LIBRARY=$(realpath $2)
O=$R/gpurun_out/prof_ode2
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH"
P3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"
k=0
for P in "$P1" "$P2" "$P3"; do
  k=$((k+1))
  rocprofv3 --pmc $P --kernel-trace -d $O -o p$k --output-format csv -- python3 $R/tools/bench_kernels.py --n 256 --reps 3 --only "ode_step $WHAT" > $O/p$k.log 2>&1 || { echo "pass $k failed"; tail -5 $O/p$k.log; }
done
cd $R
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_ode2/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if n.startswith('ode_step_kernel'):
            agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_ode2/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if n.startswith('ode_step_kernel'):
            dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    if dur[k]:
        ms = sorted(dur[k])[len(dur[k]) // 2]
        print('   median duration under the counters %.3f ms' % ms)
        if 'GRBM_GUI_ACTIVE' in m:
            print('   clock %.2f GHz' % (m['GRBM_GUI_ACTIVE'] / 8 / ms / 1e6))
    for n in sorted(m):
        print('   %-28s %.4g' % (n, m[n]))
    if 'GRBM_GUI_ACTIVE' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print('   kernel cycles (per XCD) %.4g; per SIMD: VALU active %.3f' % (cyc, 4 * m.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * cyc)))
PY
rm -f $O/*counter_collection.csv $O/*kernel_trace.csv
