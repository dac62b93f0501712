#!/bin/bash
# round 6: non-temporal state rows in the CLASS kernel too (cnt): one process (TP06 512^3, ToR-ORd 256^3), then the 401^3 shell
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 6 --allocs 1 $L/libbeat_hip.so $L/libbeat_hip_cnt.so 2>&1 | grep classes | tee gpurun_out/r06_inproc_cls_nt.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 8 --allocs 1 --dt 0.05 $L/libbeat_hip.so $L/libbeat_hip_cnt.so 2>&1 | grep classes | tee -a gpurun_out/r06_inproc_cls_nt.txt
shell() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in cur cur cnt cnt cur cnt; do
  if [ $x = cur ]; then shell cur libbeat_hip.so; else shell cnt libbeat_hip_cnt.so; fi
done | tee -a gpurun_out/r06_inproc_cls_nt.txt
