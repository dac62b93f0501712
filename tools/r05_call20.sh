#!/bin/bash
# round 5: the new default-route test; then one rank's share of the N = 8 decomposition (512 x 512 x 64): bench line + kernel trace with the
# idle analysis, fused single-slab route and the in-library decomposed loop on a one-rank communicator
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_var_gpu.py -x -q -m gpu -k "default_route" > gpurun_out/r05_tests20.log 2>&1; echo "default-route test rc $?"; tail -3 gpurun_out/r05_tests20.log
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0
for mode in fused dist; do
  if [ $mode = dist ]; then export BEAT_FORCE_DISTRIBUTED=1; else unset BEAT_FORCE_DISTRIBUTED; fi
  python3 $R/bench.py --size 512 --size-z 64 --steps 50 --warmup 10 --no-front --cpu-sample 0 > $R/gpurun_out/r05_slab64_$mode.json 2> $R/gpurun_out/r05_slab64_$mode.err || echo "bench $mode failed"
  python3 -c "
import json;d=json.loads(open('$R/gpurun_out/r05_slab64_$mode.json').read().strip().splitlines()[-1]);print('$mode', round(d['ms_per_step'],3),'ode',round(d['config']['ode_ms'],3),'pde',round(d['config']['pde_ms'],3),'k',d['config']['pcg_iterations_per_step'])"
  rm -rf /tmp/tr_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr_$mode -o t --output-format csv -- python3 $R/bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --no-front --cpu-sample 0 > /tmp/tr_$mode.json 2> /tmp/tr_$mode.err || echo "trace $mode failed"
  python3 $R/tools/trace_gaps.py /tmp/tr_$mode --last 6
  python3 - /tmp/tr_$mode <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ion = [i for i, r in enumerate(rows) if "ode_step_kernel<" in r[2]]
a, b = ion[-12], ion[-2]
d = collections.defaultdict(list)
for s, e, k in rows[a:b]:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    d[k[:44]].append(e - s)
n = 10
print("   span per step %.3f ms" % ((rows[b][0] - rows[a][0]) / n / 1e6))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("   %-44s n/step %5.2f  mean %7.1f us  per step %7.1f us" % (k, len(v) / n, sum(v) / len(v) / 1e3, sum(v) / n / 1e3))
PY
done 2>&1 | tee $R/gpurun_out/r05_slab64_trace.txt
