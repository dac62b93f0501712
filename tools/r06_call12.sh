#!/bin/bash
# round 6: the GRL1 increment of TP06's seven non-gate states as f dt phi(J dt) (the build) against the literal expression (phi0):
# TP06 tests, in one process on the same memory, then bench processes (each build twice in a row)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_golden_gpu.py tests/test_soak_gpu.py -x -q -m gpu > gpurun_out/r06_tests12.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests12.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 $L/libbeat_hip_phi0.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_phi.txt
run() { BEAT_HIP_LIBRARY=$L/$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'ode', round(f['ode_ms'],3), '| place', c.get('state_placement')['candidates'])"; }
for x in phi0 phi0 phi phi phi0 phi phi0 phi; do
  if [ $x = phi0 ]; then run phi0 libbeat_hip_phi0.so; else run phi libbeat_hip.so; fi
done | tee gpurun_out/r06_ab_phi.txt
