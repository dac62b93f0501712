#!/bin/bash
# round 6: with one block per tile as the default -- non-temporal loads / stores of the state rows (nt1 / nt2 / nt3) and the step without
# scheduling fences (nf), in one process and in the 512^3 bench (each build twice in a row); then the 401^3 shell
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 500 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 --json gpurun_out/r06_inproc_tp06_nt.json $L/libbeat_hip.so $L/libbeat_hip_nt1.so $L/libbeat_hip_nt2.so $L/libbeat_hip_nt3.so $L/libbeat_hip_nf.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_nt.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 8 --allocs 1 --dt 0.05 $L/libbeat_hip.so $L/libbeat_hip_nt1.so $L/libbeat_hip_nt2.so $L/libbeat_hip_nt3.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_torord_nt.txt
run() { BEAT_HIP_LIBRARY=$L/$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'ode', round(f.get('ode_ms', 0),3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for x in cur cur nt3 nt3 nf nf cur nt3 nf cur nt3 nf; do
  case $x in cur) run cur libbeat_hip.so;; nt3) run nt3 libbeat_hip_nt3.so;; nf) run nf libbeat_hip_nf.so;; esac
done | tee gpurun_out/r06_ab_nt.txt
shell() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in cur cur nt3 nt3 cur nt3; do
  case $x in cur) shell cur libbeat_hip.so;; nt3) shell nt3 libbeat_hip_nt3.so;; esac
done | tee gpurun_out/r06_ab_nt_shell.txt
