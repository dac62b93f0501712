#!/bin/bash
# round 6: non-temporal loads / stores in the register-row PCG kernels (rn1 loads, rn2 stores, rn3 both) against the build; bench processes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
run() { BEAT_HIP_LIBRARY=$L/$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'ode', round(f['ode_ms'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2; do
  run cur libbeat_hip.so; run rn1 libbeat_hip_rn1.so; run rn2 libbeat_hip_rn2.so; run rn3 libbeat_hip_rn3.so
done | tee gpurun_out/r06_ab_rr_nt.txt
