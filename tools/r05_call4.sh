#!/bin/bash
# round 5, fourth GPU call: right-hand-side kernel without spilled SGPRs (A = library at the previous commit), then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
L=$PWD/fenicsx-beat_amd/beat/lib
run() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for x in A B A B A B A B; do
  if [ $x = A ]; then run A libbeat_hip_base.so; else run B libbeat_hip.so; fi
done | tee gpurun_out/r05_ab_rr.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full2.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r05_full2.log
