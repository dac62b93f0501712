#!/bin/bash
# round 6: blocks per ionic launch with 4 (TP06) / 3 (ToR-ORd) waves per SIMD: one block per tile against the tile loop, in one process
# and in the 512^3 bench / the 401^3 shell (process by process, each setting twice in a row)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 500 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 --json gpurun_out/r06_inproc_tp06_grid.json $L/libbeat_hip.so $L/libbeat_hip.so@BEAT_ODE_GRID=0 $L/libbeat_hip.so@BEAT_ODE_GRID=262144 $L/libbeat_hip.so@BEAT_ODE_GRID=131072 $L/libbeat_hip.so@BEAT_ODE_GRID=65536 $L/libbeat_hip_nf.so@BEAT_ODE_GRID=0 $L/libbeat_hip_nt3.so@BEAT_ODE_GRID=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_grid.txt
run() { BEAT_ODE_GRID=$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'ode', round(f.get('ode_ms', 0),3), 'pde', round(f['pde_ms'],3), 'k', f['pcg_iterations_per_step'])"; }
for x in loop loop tile tile loop tile loop tile; do
  if [ $x = loop ]; then run loop 24576; else run tile 0; fi
done | tee gpurun_out/r06_ab_grid.txt
shell() { BEAT_ODE_GRID=$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in loop loop tile tile loop tile; do
  if [ $x = loop ]; then shell loop 24576; else shell tile 0; fi
done | tee gpurun_out/r06_ab_grid_shell.txt
