#!/bin/bash
# A/B of one environment switch on the headline bench, alternating within ONE gpurun call (one box):
#   bash tools/ab_env.sh NAME A B [reps] [bench flags...]     -> lines "NAME=value ms_per_step ode_ms pde_ms" in run order
# (the first run on a fresh box is part of what this shows: keep the order in the record)
NAME=$1; A=$2; B=$3; REPS=${4:-3}; shift 4
for i in $(seq 1 $REPS); do
  for v in $A $B; do
    env $NAME=$v python3 bench.py --no-front --cpu-sample 0 "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$NAME=$v', round(r['ms_per_step'], 3), 'ode', round(r['config']['ode_ms'], 3), 'pde', round(r['config']['pde_ms'], 3), 'host', round(r['ms_per_step'] - r['config']['ode_ms'] - r['config']['pde_ms'], 3), 'k', r['config']['pcg_iterations_per_step'], 'sclk', (r['config'].get('clocks') or {}).get('sclk_mhz'), 'W', (r['config'].get('clocks') or {}).get('power_w'), 'Tj', (r['config'].get('clocks') or {}).get('junction_c'), flush=True)"
  done
done
