#!/bin/bash
# round 6: rows per wave x planes ahead of the register-row kernels at 512^3 once more (BEAT_RR_RY x BEAT_RR_PD)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
run() { env "$@" BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$*', '|', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2; do
run X=0
run BEAT_RR_RY=2
run BEAT_RR_RY=2 BEAT_RR_PD=2
run BEAT_RR_RY=2 BEAT_RR_PD=3
run BEAT_RR_RY=2 BEAT_RR_BLOCKS=8192
run BEAT_RR_RY=2 BEAT_RR_PD=2 BEAT_RR_BLOCKS=8192
done | tee gpurun_out/r06_rr_sweep2.txt
