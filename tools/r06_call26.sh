#!/bin/bash
# round 6: the PCG's work fields placed by default: GPU suite, then the shell and the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests26.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r06_tests26.log
[ $rc = 0 ] || exit 1
shell() { BEAT_WORK_PLACE=$2 timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>/dev/null | tail -1 | sed "s/^/$1 /"; }
for x in one one three three one three; do
  if [ $x = one ]; then shell one 1; else shell three 3; fi
done | tee gpurun_out/r06_shell_work_place.txt
timeout -k 10 400 python bench.py > gpurun_out/r06_bench26.json 2> gpurun_out/r06_bench26.err; python -c "
import json;d=json.loads(open('gpurun_out/r06_bench26.json').read().strip().splitlines()[-1]);c=d['config'];print(d['ms_per_step'], c['ode_ms'], c['pde_ms'], d['roofline']['frac'], c['state_placement'], c['work_placement'], d['developed_front']['ms_per_step'], d['batched_solve']['ms_per_step'])"
