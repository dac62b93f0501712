#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 $L/libbeat_hip.so $L/libbeat_hip_gphi.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_gphi.txt
