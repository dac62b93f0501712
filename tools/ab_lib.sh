#!/bin/bash
# Paired A/B of two builds of the library on one box: bench.py (512^3) A A B B A B, printing ms/step and the ionic time.
# usage: bash tools/ab_lib.sh <alternative .so>   (A = the in-tree library, B = the alternative)
ALT=$1
run() { BEAT_HIP_LIBRARY=$2 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'])"; }
A=$PWD/fenicsx-beat_amd/beat/lib/libbeat_hip.so
for x in A A B B A B A B; do if [ $x = A ]; then run A $A; else run B $ALT; fi; done
