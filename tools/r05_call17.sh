#!/bin/bash
# round 5: generated models (rows / classes / run); the whole GPU suite; then the register-row kernels with the four waves of a block
# on four adjacent row blocks (BEAT_RR_BY_ROWS bit per MODE: 1 PDOT, 2 RUPD, 4 RHS) against the x-segment-fastest mapping
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ode_file_gpu.py -x -q -m gpu > gpurun_out/r05_tests16a.log 2>&1; rc=$?; echo "ode_file rc $rc"; tail -3 gpurun_out/r05_tests16a.log
run() { BEAT_BENCH_BATCHED=0 BEAT_RR_BY_ROWS=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for rep in 1 2 3; do
  run map0 0
  run rhs4 4
  run all7 7
done | tee gpurun_out/r05_ab_rr_by_rows.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full7.log 2>&1; echo "full rc $?"; tail -5 gpurun_out/r05_full7.log
