#!/bin/bash
# Row-stride skew of the state array (BEAT_STATE_SKEW doubles added to ld) against the ionic kernels and their
# memory-only probe build: [N=512 ONLY="ode_step tp06"] bash tools/skew_probe.sh <probe1.so>
for skew in ${SKEWS:-0 544 2080 8224 33824}; do
  for lib in fenicsx-beat_amd/beat/lib/libbeat_hip.so "$@"; do
    echo "== skew $skew $(basename $lib)"
    BEAT_STATE_SKEW=$skew BEAT_HIP_LIBRARY=$(realpath $lib) python3 tools/bench_kernels.py --n ${N:-256} --reps 5 --only "${ONLY:-ode_step}" 2>/dev/null
  done
done
