#!/bin/bash
# round 5: with aligned segments, rows per wave (BEAT_RR_RY) and blocks per launch (BEAT_RR_BLOCKS) again: the 512 x 512 x 64 slab, 256^3 iso, 512^3
set -o pipefail
mkdir -p gpurun_out
export BEAT_BENCH_BATCHED=0
run() { env $1 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 [$2]', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'])"; }
for rep in 1 2; do
  for e in BEAT_RR_RY=2 BEAT_RR_RY=4; do run $e "--size 512 --size-z 64 --steps 50 --warmup 10"; done
  for e in BEAT_RR_RY=2 BEAT_RR_RY=4; do run $e "--size 256 --iso --steps 100 --warmup 20"; done
  for e in BEAT_RR_BLOCKS=4096 BEAT_RR_BLOCKS=8192 BEAT_RR_BLOCKS=2048; do run $e "--steps 20 --warmup 5"; done
done | tee gpurun_out/r05_rr_align_sweep.txt
