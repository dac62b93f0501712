#!/bin/bash
# Ionic-kernel time against the number of blocks per launch (BEAT_ODE_GRID; a block walks over tiles of 256 nodes).
# usage: bash tools/ode_grid_sweep.sh <size> <grid> [<grid> ...]
S=$1; shift
for g in "$@" "$@"; do
  BEAT_ODE_GRID=$g python bench.py --size $S --iso --steps 100 --warmup 10 --cpu-sample 0 --no-front 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('grid', $g, 'ms/step', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],4), 'pde', round(d['config']['pde_ms'],4))"
done
