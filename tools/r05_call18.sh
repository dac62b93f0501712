#!/bin/bash
# round 5: (1) the from_ode GPU tests; (2) HBM read traffic (FETCH_SIZE pass) and times (kernel-trace pass) of the register-row kernels
# with the x-segment-fastest wave mapping (BEAT_RR_BY_ROWS=0) and with the four waves of a block on adjacent row blocks (=7)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_ode_file_gpu.py -x -q -m gpu > gpurun_out/r05_tests18.log 2>&1; echo "ode_file rc $?"; tail -3 gpurun_out/r05_tests18.log
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0
for m in 0 7; do
  export BEAT_RR_BY_ROWS=$m
  rm -rf /tmp/pm_$m /tmp/kt_$m
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pm_$m -o pmc --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-front > /tmp/pm_$m.json 2> /tmp/pm_$m.err || echo "pmc pass failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$m -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-front > /tmp/kt_$m.json 2> /tmp/kt_$m.err || echo "trace pass failed"
  echo "== BEAT_RR_BY_ROWS=$m"
  python3 - /tmp/pm_$m /tmp/kt_$m <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rr_kernel" in n and r["Counter_Name"] == "FETCH_SIZE":
            acc[n[n.index("rr_kernel"):][:28]].append(float(r["Counter_Value"]))
for n, v in sorted(acc.items()):
    v = [x for x in v if x > 1024.0]
    if v:
        print("   %-28s launches %3d  read %.3f GiB (2 x FETCH_SIZE)" % (n, len(v), 2 * sum(v) / len(v) / 2**20))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rr_kernel" in n:
            dur[n[n.index("rr_kernel"):][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(dur.items()):
    v = [x for x in v if x > 50.0]
    if v:
        print("   %-28s launches %3d  mean %.1f us  min %.1f" % (n, len(v), sum(v) / len(v), min(v)))
PY
done 2>&1 | tee $R/gpurun_out/r05_rr_by_rows_pmc.txt
