#!/bin/bash
# round 6: the guess history placed too (BEAT_HIST_PLACE=3 against 1); guess tests, then bench processes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_guess_gpu.py tests/test_properties_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { BEAT_HIST_PLACE=$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for x in one one three three one three one three one three; do
  if [ $x = one ]; then run one 1; else run three 3; fi
done | tee gpurun_out/r06_ab_hist_place.txt
