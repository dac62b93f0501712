#!/bin/bash
# round 6: non-temporal stores of p / q in the per-node tile kernel (vn2) on the 401^3 shell, process by process
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
shell() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in cur cur vn2 vn2 cur vn2 cur vn2; do
  if [ $x = cur ]; then shell cur libbeat_hip.so; else shell vn2 libbeat_hip_vn2.so; fi
done | tee gpurun_out/r06_ab_vtl_nt.txt
