#!/bin/bash
# round 5: bench.py as the driver runs it (front regime included) under the kernel trace: the step() part against the library-loop part,
# per-kernel means and idle; then the same without the front regime
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in front nofront; do
  rm -rf /tmp/trace_$mode
  extra=""; [ $mode = nofront ] && extra="--no-front"
  timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/trace_$mode -o t --output-format csv -- python3 $R/bench.py --cpu-sample 0 $extra > $R/gpurun_out/r05_trace_$mode.json 2> $R/gpurun_out/r05_trace_$mode.err || echo "profile failed"
  python3 -c "
import json;d=json.loads(open('$R/gpurun_out/r05_trace_$mode.json').read().strip().splitlines()[-1]);b=d['batched_solve'];print('$mode: step', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), '| batched', round(b['ms_per_step'],3), 'ode', round(b['ode_ms'],3))"
  python3 - /tmp/trace_$mode <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ion = [i for i, r in enumerate(rows) if "ode_step_kernel<" in r[2]]
print(len(ion), "ionic launches")
def part(a, b, tag):
    seg = rows[ion[a]:ion[b]]
    d = collections.defaultdict(list)
    for s, e, k in seg:
        d[k.replace("(anonymous namespace)::", "").replace("void ", "")[:40]].append(e - s)
    span = rows[ion[b]][0] - rows[ion[a]][0]
    busy = sum(e - s for s, e, _ in seg)
    print(tag, "span per step %.3f ms, busy %.3f, idle %.1f us" % (span / (b - a) / 1e6, busy / (b - a) / 1e6, (span - busy) / (b - a) / 1e3))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("   %-40s n/step %5.2f  mean %8.1f us  per step %8.1f us" % (k, len(v) / (b - a), sum(v) / len(v) / 1e3, sum(v) / (b - a) / 1e3))
# the timed step() part: launches 5..24 (5 warm-up + 20 timed); the library loop's timed part: the last 20 launches
part(6, 24, "step() calls")
part(len(ion) - 19, len(ion) - 1, "library loop")
PY
done 2>&1 | tee $R/gpurun_out/r05_batched_vs_step.txt
