#!/bin/bash
# round 5, eighth GPU call: kernel trace of the step loop with and without the open (lazy) solve: where does the device idle?
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf /tmp/trace_$v
  BEAT_LAZY_KSP=$v BEAT_BENCH_BATCHED=0 timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/trace_$v -o t --output-format csv -- python3 $R/bench.py --cpu-sample 0 --no-front --steps 12 --warmup 4 > $R/gpurun_out/r05_trace_lazy$v.json 2> $R/gpurun_out/r05_trace_lazy$v.err || echo "profile failed"
  python3 $R/tools/trace_gaps.py /tmp/trace_$v --last 10 | tee $R/gpurun_out/r05_trace_gaps_lazy$v.txt
done
