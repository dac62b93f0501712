#!/bin/bash
# round 5: where the decomposed loop's remaining idle time sits (one rank's 512 x 512 x 64 share, one-rank RCCL communicator): gaps by
# kernel boundary, for the default transport, the serial ordering (one stream) and the mailbox transport
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0 BEAT_FORCE_DISTRIBUTED=1 BEAT_BENCH_ALT=0
for mode in default serial; do
  if [ $mode = serial ]; then export BEAT_DIST_SERIAL=1; else unset BEAT_DIST_SERIAL; fi
  rm -rf /tmp/tr_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tr_$mode -o t --output-format csv -- python3 $R/bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --no-front --cpu-sample 0 > /tmp/tr_$mode.json 2> /tmp/tr_$mode.err || { echo "trace $mode failed"; tail -3 /tmp/tr_$mode.err; }
  echo "== $mode"; python3 $R/tools/trace_gaps.py /tmp/tr_$mode --last 8 --pairs | tail -26
  ls /tmp/tr_$mode/*/ | head -5
done 2>&1 | tee $R/gpurun_out/r05_slab64_gap_pairs.txt
