#!/bin/bash
# round 6: does the level of the 19-row pattern belong to the allocation or to the process?  (tools/place_probe3.py, separate processes)
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do echo "== process $i"; timeout -k 10 200 python tools/place_probe3.py --allocs 6 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_place_probe3.txt
