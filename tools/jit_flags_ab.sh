#!/bin/bash
# Compiler flags against the ionic kernels, without rebuilding the library: the sparse-rows instance the library compiles at run time
# (csrc/beat_ode_jit.h; with one varying conductance it is the uniform kernel plus one row load) built with extra flags
# (BEAT_JIT_EXTRA_FLAGS, part of the cache key) and timed by tools/bench_param_classes.py.   bash tools/jit_flags_ab.sh [tp06|torord]
MODEL=${1:-torord}
run() {
  BEAT_JIT_EXTRA_FLAGS="$1" python3 tools/bench_param_classes.py --model $MODEL --reps 12 2>/dev/null | grep -E "^uniform|compiled instance " | sed 's/ (.*//' | tr '\n' '|'
  echo "  <= [$1]"
}
run ""
run "-mllvm -amdgpu-sched-strategy=gcn-max-ilp"
run "-mllvm -amdgpu-sched-strategy=gcn-max-memory-clause"
run "-mllvm -amdgpu-use-amdgpu-trackers"
run "-mllvm -amdgpu-schedule-metric-bias=100"
run "-mllvm -amdgpu-schedule-metric-bias=0"
run "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule"
run "-mllvm -amdgpu-max-memory-clause=4"
run ""
