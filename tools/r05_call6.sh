#!/bin/bash
# round 5, sixth GPU call: ring of 12 on the per-node path (A = BEAT_VAR_RING=6), then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
for v in 12 6 12 6; do
  echo "BEAT_VAR_RING=$v"; BEAT_VAR_RING=$v timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done | tee gpurun_out/r05_biv400_ring.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full3.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r05_full3.log
