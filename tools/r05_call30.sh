#!/bin/bash
# round 5: the count of the open solve from LDS (once per block) against the scalar load per tile: bench A/B on one box
set -o pipefail
mkdir -p gpurun_out
L=$PWD/fenicsx-beat_amd/beat/lib
timeout -k 10 300 env BEAT_HIP_LIBRARY=$L/libbeat_hip_lds.so python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "leaves_its_solve_open or fused_step_equals or deferred" 2>&1 | tail -2
run() { BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());b=d['batched_solve'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), '| batched', round(b['ms_per_step'],3), 'ode', round(b.get('ode_ms'),3))"; }
for rep in 1 2 3 4 5 6 7 8; do
  run scalar $L/libbeat_hip.so
  run lds $L/libbeat_hip_lds.so
done | tee gpurun_out/r05_ab_pending_lds.txt
