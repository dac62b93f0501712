#!/bin/bash
# round 6: the register-row kernels' launch switches once more, with non-temporal stores in (BEAT_RR_BLOCKS / _PD / _GUESS_RY / _BY_ROWS)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
run() { env "$@" BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$*', '|', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2; do
run X=0
run BEAT_RR_BLOCKS=2048
run BEAT_RR_BLOCKS=8192
run BEAT_RR_BLOCKS=16384
run BEAT_RR_PD=2
run BEAT_RR_GUESS_RY=4
run BEAT_RR_BY_ROWS=4
run BEAT_RR_BY_ROWS=7
done | tee gpurun_out/r06_rr_sweep.txt
