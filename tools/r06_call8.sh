#!/bin/bash
# round 6: what differs between the two levels consecutive processes alternate between?  Six bench processes of the build, the clocks
# / power record of each beside its times
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3 4 5 6; do
  BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config']['clocks'];f=d['developed_front'];print('run $i', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), '| front', round(f['ms_per_step'],3), '| sclk', c['sclk_mhz'], 'power', c['power_w'], 'T', c['junction_c'], c['hbm_c'], '| stream', round(d['roofline']['inplace_stream']['rate']), round(d['roofline']['inplace_stream']['rows_pattern']['rate']))"
done | tee gpurun_out/r06_levels.txt
