#!/bin/bash
# round 5, last measurement pass: the long run through the public API with open solves, then the shell under rocprofv3 and the shell at
# BASELINE configs[4]'s own size
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/soak_api.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_soak_api.txt
rm -rf gpurun_out/prof_shell
bash tools/profile_shell.sh 400
python3 tools/summarize_prof.py gpurun_out/prof_shell gpurun_out/prof_shell/summary.md "Round 5: voxel shell (401^3 box) under rocprofv3" --json gpurun_out/prof_shell/summary.json > /dev/null
rm -f gpurun_out/prof_shell/*kernel_trace.csv gpurun_out/prof_shell/*agent_info.csv gpurun_out/prof_shell/*counter_collection.csv
for i in 1 2; do timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1; done | tee gpurun_out/r05_biv400_final2.txt
timeout -k 10 500 python tools/bench_biv.py --size 520 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r05_biv520_final2.txt
