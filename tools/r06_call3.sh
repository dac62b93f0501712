#!/bin/bash
# round 6: the ionic kernels of the old and the new addressing on the SAME device memory, one process (tools/ab_ode_inproc.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 12 --allocs 3 --json gpurun_out/r06_inproc_tp06.json $L/libbeat_hip_old.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 12 --allocs 2 --dt 0.05 --json gpurun_out/r06_inproc_torord.json $L/libbeat_hip_old.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_torord.txt
