#!/bin/bash
# round 5: the open solve's update count read once per launch (in-tree) against once per tile (libbeat_hip_pertile.so): bench A/B on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "leaves_its_solve_open or batched_solve or fused or deferred" > gpurun_out/r05_tests14.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05_tests14.log
A=$PWD/fenicsx-beat_amd/beat/lib/libbeat_hip.so
B=$PWD/fenicsx-beat_amd/beat/lib/libbeat_hip_pertile.so
run() { BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());b=d['batched_solve'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| batched', round(b['ms_per_step'],3), 'ode', b.get('ode_ms'), 'frac', round(d['roofline']['frac'],3))"; }
for x in once tile once tile once tile once tile; do if [ $x = once ]; then run once $A; else run tile $B; fi; done | tee gpurun_out/r05_ab_pending_read.txt
for v in A B A B; do
  if [ $v = A ]; then L=$A; else L=$B; fi
  echo "lib $v"; BEAT_HIP_LIBRARY=$L timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done | tee gpurun_out/r05_biv400_pending_read.txt
