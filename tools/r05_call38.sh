#!/bin/bash
# round 5: the whole GPU suite on the build with raw-buffer loads in the register-row kernels, then their times and read traffic
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full10.log 2>&1; echo "full rc $?"; tail -4 gpurun_out/r05_full10.log
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0
rm -rf /tmp/pm /tmp/kt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pm -o pmc --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-front > /tmp/pm.json 2> /tmp/pm.err || echo "pmc pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-front > /tmp/kt.json 2> /tmp/kt.err || echo "trace pass failed"
python3 - /tmp/pm /tmp/kt <<'PY' | tee $R/gpurun_out/r05_rr_buf_kernels.txt
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
