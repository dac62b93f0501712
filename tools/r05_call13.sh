#!/bin/bash
# round 5: (1) kernel trace of bench.py including the library's own step loop (how much does THAT loop idle, which kernels differ);
# (2) bench.py --gpus 4 as the driver launches it, four ranks sharing this box's GPU over gloo: the line with multi_rank_parity.
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
BEAT_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 4 --steps 6 --warmup 2 --size 128 --cpu-sample 0 > gpurun_out/r05_rehearsal4.json 2> gpurun_out/r05_rehearsal4.err || { echo "rehearsal failed"; tail -5 gpurun_out/r05_rehearsal4.err; }
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r05_rehearsal4.json").read().strip().splitlines()[-1])
print("n_gpus", r["n_gpus"], "ms/step", r["ms_per_step"], json.dumps(r.get("multi_rank_parity"))[:1500])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_b
BEAT_BENCH_BATCHED_EVENTS=0 timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/trace_b -o t --output-format csv -- python3 $R/bench.py --cpu-sample 0 --no-front --steps 12 --warmup 4 > $R/gpurun_out/r05_trace_batched.json 2> $R/gpurun_out/r05_trace_batched.err || echo "profile failed"
python3 $R/tools/trace_gaps.py /tmp/trace_b --last 40 > $R/gpurun_out/r05_trace_gaps_batched.txt
python3 - <<'PY'
# per-kernel mean duration in the step() part and in the batched part of the trace
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/trace_b/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ion = [i for i, r in enumerate(rows) if "ode_step_kernel<" in r[2]]
print(len(ion), "ionic launches")
def part(a, b, tag):
    seg = rows[ion[a]:ion[b]]
    d = collections.defaultdict(list)
    for s, e, k in seg:
        d[k[:60]].append(e - s)
    span = rows[ion[b]][0] - rows[ion[a]][0]
    print(tag, "span per step %.3f ms" % (span / (b - a) / 1e6))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("   %-60s n/step %5.2f  mean %8.1f us  per step %8.1f us" % (k, len(v) / (b - a), sum(v) / len(v) / 1e3, sum(v) / (b - a) / 1e3))
if len(ion) >= 32:
    part(5, 15, "step() calls")
    part(len(ion) - 11, len(ion) - 1, "library loop")
PY
cat $R/gpurun_out/r05_trace_gaps_batched.txt | tail -14
