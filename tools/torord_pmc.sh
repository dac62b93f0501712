#!/bin/bash
# SQ / GRBM counters of the ionic kernels at 256^3 (tools/bench_kernels.py): VALU instructions per wave, VALU-busy
# fraction, parked wave cycles.  Output printed; raw files under gpurun_out/prof_ode256/.
R=$PWD
O=$R/gpurun_out/prof_ode256
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O -o sq --output-format csv -- python3 $R/tools/bench_kernels.py --n 256 --reps 3 > $O/sq.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/prof_ode256/sq_counter_collection.csv')):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if n.startswith('ode_step_kernel'):
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    nodes = 256 ** 3
    print(k)
    print('   VALU instr / node %.0f   VALU busy %.2f   parked %.2f   clock %.2f GHz (if the launch took GUI/8 cycles)' % (
        m['SQ_INSTS_VALU'] * 64 / nodes, 4 * m['SQ_ACTIVE_INST_VALU'] / (1024 * m['GRBM_GUI_ACTIVE'] / 8), m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 0.0))
    print('   GUI cycles per XCD %.3g, waves %.0f' % (m['GRBM_GUI_ACTIVE'] / 8, m['SQ_WAVES']))
PY
grep -E "ode_step" $O/sq.log
rm -f $O/*counter_collection.csv $O/*kernel_trace.csv
