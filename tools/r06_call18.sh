#!/bin/bash
# round 6: TP06's exp() with 2^m added into the exponent field (the build) against shift + v_ldexp_f64 (ei0): the GPU suite, one process,
# then bench processes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests18.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests18.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 $L/libbeat_hip_ei0.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_expint.txt
run() { BEAT_HIP_LIBRARY=$L/$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'ode', round(f['ode_ms'],3), '| place', c.get('state_placement')['candidates'])"; }
for x in ldexp ldexp int int ldexp int; do
  if [ $x = ldexp ]; then run ldexp libbeat_hip_ei0.so; else run int libbeat_hip.so; fi
done | tee gpurun_out/r06_ab_expint.txt
