#!/bin/bash
# round 5: every demo script as its own fresh process (what a user does), small sizes
set -o pipefail
mkdir -p gpurun_out /tmp/demo_out
cd demos
for cmd in "diffusion.py --n 20 --T 2.5 --dt 0.1" "simple_ode.py --cells 16 --T 420" "fitzhughnagumo.py" "slab_ecg.py --dx 0.5 --T 6 --beats 1 --out /tmp/demo_out/ecg" "pace_train.py --dx 0.5 --s1 1 --bcl 320" "niederer_benchmark.py --dx 0.5 --dt 0.05" "ode_file_slab.py --dx 0.5 --T 5 --beats 1"; do
  echo "== $cmd"
  timeout -k 10 280 python3 $cmd 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-220
  echo "rc ${PIPESTATUS[0]}"
done 2>&1 | tee ../gpurun_out/r05_demos_fresh.txt
