#!/bin/bash
# round 5: kernel trace of bench.py with the library's own step loop at the end (batched_solve): how much does THAT loop idle?
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/trace_b
BEAT_BENCH_BATCHED_EVENTS=0 timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/trace_b -o t --output-format csv -- python3 $R/bench.py --cpu-sample 0 --no-front --steps 12 --warmup 4 > $R/gpurun_out/r05_trace_batched.json 2> $R/gpurun_out/r05_trace_batched.err || echo "profile failed"
python3 $R/tools/trace_gaps.py /tmp/trace_b --last 40 | tee $R/gpurun_out/r05_trace_gaps_batched.txt
