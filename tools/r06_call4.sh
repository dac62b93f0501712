#!/bin/bash
# round 6: ionic kernels with saddr addressing + per-tile kernel arguments + the class kernel's per-lane state in LDS (TP06 4 waves, ToR-ORd
# 3 waves per SIMD, class kernels included): GPU suite, in-process A/B against the library of the previous commit (libbeat_hip_base.so),
# then the benches process by process (base base new new base new base new: consecutive processes alternate between two levels)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests4.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests4.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 10 --allocs 2 --json gpurun_out/r06_inproc_tp06.json $L/libbeat_hip_base.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 10 --allocs 2 --dt 0.05 --json gpurun_out/r06_inproc_torord.json $L/libbeat_hip_base.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_torord.txt
run() { BEAT_BENCH_BATCHED=0 BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'ode', round(f.get('ode_ms', 0),3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for x in base base new new base new base new; do
  if [ $x = base ]; then run base $L/libbeat_hip_base.so; else run new $L/libbeat_hip.so; fi
done | tee gpurun_out/r06_ab_saddr.txt
shell() { BEAT_HIP_LIBRARY=$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in base base new new base new; do
  if [ $x = base ]; then shell base $L/libbeat_hip_base.so; else shell new $L/libbeat_hip.so; fi
done | tee gpurun_out/r06_ab_saddr_shell.txt
