#!/bin/bash
# round 6: the build (saddr addressing, one block per tile, non-temporal rows, placement of the state array chosen from three
# candidates): GPU suite, then six bench processes with and without the placement choice (alternating in pairs)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests10.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests10.log
[ $rc = 0 ] || exit 1
run() { BEAT_STATE_PLACE=$2 BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'ode', round(f['ode_ms'],3), '| place', c.get('state_placement'), '| rows', round(d['roofline']['inplace_stream']['rows_pattern']['rate']))"; }
for x in one one three three one three one three; do
  if [ $x = one ]; then run one 1; else run three 3; fi
done | tee gpurun_out/r06_ab_place.txt
