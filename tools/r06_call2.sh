#!/bin/bash
# round 6: the ionic kernels with rows addressed as uniform tile base + ONE 32-bit lane offset (saddr form; TP06 134 -> 118 VGPRs = 4 waves
# per SIMD, ToR-ORd 216 -> 164 = 3 waves): the GPU suite on the build, then bench A/B against the old addressing (libbeat_hip_old.so),
# then the voxel shell (ToR-ORd classes) on old / new / new with the class kernel forced to 3 waves (libbeat_hip_tw3.so)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests2.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests2.log
[ $rc = 0 ] || exit 1
run() { BEAT_BENCH_BATCHED=0 BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'ode', round(f.get('ode_ms', 0),3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for rep in 1 2 3 4; do
  run old $L/libbeat_hip_old.so
  run new $L/libbeat_hip.so
done | tee gpurun_out/r06_ab_saddr.txt
shell() { BEAT_HIP_LIBRARY=$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -3 | sed "s/^/$1 /"; }
for rep in 1 2; do
  shell old $L/libbeat_hip_old.so
  shell new $L/libbeat_hip.so
  shell tw3 $L/libbeat_hip_tw3.so
done | tee gpurun_out/r06_ab_saddr_shell.txt
