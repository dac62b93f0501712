#!/bin/bash
# Paired runs of bench.py (512^3, 20 steps) over settings given as "NAME ENV=VAL ..." strings:
#   bash tools/ab_bench.sh "skew0 BEAT_STATE_SKEW=0" "auto" "base BEAT_HIP_LIBRARY=$PWD/fenicsx-beat_amd/beat/lib/alt/lib_base.so"
for round in 1 2 3; do
  for spec in "$@"; do
    name=${spec%% *}
    envs=""
    [ "$spec" != "$name" ] && envs=${spec#* }
    env $envs python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$name', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], 'frac', round(d['roofline']['frac'],3))"
  done
done
