#!/bin/bash
# round 6: tuning of the 4-wave TP06 / 3-wave ToR-ORd kernels in one process (tools/ab_ode_inproc.py): in-place opaque offsets (the build)
# against the previous commit, scheduling fences off, non-temporal rows, grid sizes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 500 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 --json gpurun_out/r06_inproc_tp06_tune.json $L/libbeat_hip_base.so $L/libbeat_hip.so $L/libbeat_hip_nf.so $L/libbeat_hip_nt3.so $L/libbeat_hip.so@BEAT_ODE_GRID=16384 $L/libbeat_hip.so@BEAT_ODE_GRID=32768 $L/libbeat_hip.so@BEAT_ODE_GRID=49152 $L/libbeat_hip.so@BEAT_ODE_GRID=8192 $L/libbeat_hip.so@BEAT_ODE_GRID=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_tune.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 8 --allocs 1 --dt 0.05 --json gpurun_out/r06_inproc_torord_tune.json $L/libbeat_hip_base.so $L/libbeat_hip.so $L/libbeat_hip_nt3.so $L/libbeat_hip.so@BEAT_ODE_GRID=12288 $L/libbeat_hip.so@BEAT_ODE_GRID=32768 $L/libbeat_hip.so@BEAT_ODE_GRID=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_torord_tune.txt
