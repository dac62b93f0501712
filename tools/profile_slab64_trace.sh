#!/bin/bash
# Kernel trace of one rank's share of the N = 8 decomposition (512 x 512 x 64 slab) through the decomposed code path on ONE GPU
# (BEAT_FORCE_DISTRIBUTED=1: a one-rank RCCL communicator): where does a step's time go -- kernels, or the gaps between them?
#   bash tools/profile_slab64_trace.sh [extra bench.py flags]
set -e
R=$PWD
O=$R/gpurun_out/prof_slab64
rm -rf $O && mkdir -p $O
export BEAT_FORCE_DISTRIBUTED=1
python3 $R/bench.py --size-z 64 --steps 30 --warmup 5 --no-front --cpu-sample 0 "$@" > $O/plain.json 2> $O/plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o trace --output-format csv -- python3 $R/bench.py --size-z 64 --steps 30 --warmup 5 --no-front --cpu-sample 0 "$@" > $O/trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections, re, json
rows = list(csv.DictReader(open('gpurun_out/prof_slab64/trace_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:52]
ode = [i for i, r in enumerate(rows) if 'ode_step_kernel' in r['Kernel_Name']]
steps = [(ode[k], ode[k + 1]) for k in range(len(ode) - 21, len(ode) - 1)]   # the last 20 whole steps
wall = busy = 0.0
per = collections.OrderedDict()
nk = 0
for a, b in steps:
    t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    wall += (t1 - t0) / 1e3
    for r in rows[a:b]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        busy += d
        nk += 1
        c = per.setdefault(name(r), [0, 0.0]); c[0] += 1; c[1] += d
n = len(steps)
print("plain run:", json.loads(open('gpurun_out/prof_slab64/plain.json').read().strip().splitlines()[-1])["ms_per_step"], "ms/step")
print("traced: %.1f us per step wall, %.1f us in kernels (%.0f %%), %.1f launches per step" % (wall / n, busy / n, 100 * busy / wall, nk / n))
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-54s %6.2f per step  %8.1f us each  %8.1f us per step" % (k, c / n, d / c, d / n))
PY
rm -f $O/*kernel_trace.csv
