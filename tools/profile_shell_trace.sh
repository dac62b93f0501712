#!/bin/bash
# Kernel trace only of the voxel-shell bench (tools/bench_biv.py): average active duration of the solver's kernels.
set -e
R=$PWD
O=$R/gpurun_out/prof_shell_trace
N=${1:-400}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o trace --output-format csv -- python3 $R/tools/bench_biv.py --n $N --steps 12 --warmup 3 > $O/trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open('gpurun_out/prof_shell_trace/trace_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = next(i for i, r in enumerate(rows) if 'ode_step_kernel' in r['Kernel_Name'])
agg = collections.OrderedDict()
for r in rows[first:]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d < 20: continue   # latched no-op launches and the scalar kernels
    c = agg.setdefault(n, []); c.append(d)
for n, l in agg.items():
    l.sort()
    print("%-46s x%5d  median %8.1f us  total %8.2f ms" % (n, len(l), l[len(l)//2], sum(l)/1e3))
PY
rm -f $O/*kernel_trace.csv
