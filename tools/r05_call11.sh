#!/bin/bash
# round 5: from_ode GPU tests + distributed tests; non-temporal coefficient loads in the tile kernel (A = shipped library)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_ode_file_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r05_tests11.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r05_tests11.log
L=$PWD/fenicsx-beat_amd/beat/lib
for x in A vnt1 A vnt1 vnt3 A vnt3; do
  if [ $x = A ]; then lib=libbeat_hip.so; else lib=libbeat_hip_$x.so; fi
  echo -n "$x: "; BEAT_HIP_LIBRARY=$L/$lib timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done | tee gpurun_out/r05_biv400_vnt.txt
