#!/bin/bash
# round 6: the 401^3 shell on one box: round 5's library, the build without / with the placement choice for the state array
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
shell() { BEAT_HIP_LIBRARY=$L/$2 BEAT_STATE_PLACE=$3 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for i in 1 2; do
  shell base libbeat_hip_base.so 1
  shell new-place1 libbeat_hip.so 1
  shell new-place3 libbeat_hip.so 3
done | tee gpurun_out/r06_shell15.txt
