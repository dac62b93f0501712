#!/bin/bash
# round 6 (experiment): the PCG's work fields placed by the best of three allocations (BEAT_WORK_PLACE=3) against one; bench processes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
run() { BEAT_WORK_PLACE=$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for x in one one three three one three one three one three; do
  if [ $x = one ]; then run one 1; else run three 3; fi
done | tee gpurun_out/r06_ab_work_place.txt
