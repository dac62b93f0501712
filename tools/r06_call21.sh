#!/bin/bash
# round 6: the build (non-temporal stores in the PCG's register-row kernels; TP06 exp back on ldexp): GPU suite, smoke(), the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests21.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests21.log
[ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 400 python bench.py > gpurun_out/r06_bench21.json 2> gpurun_out/r06_bench21.err; tail -c 300 gpurun_out/r06_bench21.err; python -c "
import json;d=json.loads(open('gpurun_out/r06_bench21.json').read().strip().splitlines()[-1]);c=d['config'];print(d['ms_per_step'], c['ode_ms'], c['pde_ms'], d['roofline']['frac'], d['roofline']['traffic_source'], d['developed_front']['ms_per_step'], d['batched_solve']['ms_per_step'], d['cpu_baseline']['value'])"
