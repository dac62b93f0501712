#!/bin/bash
# Paired timing of the ionic kernels (tools/bench_kernels.py, 256^3) over several builds of the library on one box:
#   bash tools/ab_ode.sh [alt1.so alt2.so ...]     (first = the in-tree library; the round is run twice)
L=$PWD/fenicsx-beat_amd/beat/lib/libbeat_hip.so
for round in 1 2; do
  for lib in $L "$@"; do
    echo "== $(basename $lib)"
    BEAT_HIP_LIBRARY=$(realpath $lib) python3 tools/bench_kernels.py --n ${N:-256} --reps 5 --only ode_step 2>/dev/null
  done
done
