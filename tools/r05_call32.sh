#!/bin/bash
# round 5: blocks per ionic launch (BEAT_ODE_GRID) with the final build, alternating on one box
set -o pipefail
mkdir -p gpurun_out
run() { BEAT_BENCH_BATCHED=0 BEAT_ODE_GRID=$1 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('grid $1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3))"; }
for rep in 1 2 3; do for g in 24576 16384 32768 49152 12288; do run $g; done; done | tee gpurun_out/r05_ode_grid_sweep.txt
