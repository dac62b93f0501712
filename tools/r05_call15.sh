#!/bin/bash
# round 5: how the ionic launch behind an open solve reads the solve's update count: scalar load per tile (in-tree), vector load per
# tile (pertile), once per launch (once): bench A/B/C on one box, step() and the library's loop
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "leaves_its_solve_open or batched_solve or fused or deferred" > gpurun_out/r05_tests15.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05_tests15.log
L=$PWD/fenicsx-beat_amd/beat/lib
run() { BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());b=d['batched_solve'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| batched', round(b['ms_per_step'],3), 'ode', round(b.get('ode_ms'),3), 'frac', round(d['roofline']['frac'],3))"; }
for rep in 1 2 3 4; do
  run scalar $L/libbeat_hip.so
  run tile $L/libbeat_hip_pertile.so
  run once $L/libbeat_hip_once.so
done | tee gpurun_out/r05_ab_pending_read.txt
