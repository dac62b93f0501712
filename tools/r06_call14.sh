#!/bin/bash
# round 6: the build after the instruction passes (phi polynomial for the non-gate states, exp's rounding constant pinned, no zero fill of
# the pending values; launch bounds 4 / 3 waves): GPU suite, in one process against the library of round 5's last commit, bench x 4, shell x 3
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests14.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r06_tests14.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python tools/ab_ode_inproc.py --n 512 --model tp06 --reps 8 --allocs 1 --json gpurun_out/r06_inproc_tp06_final.json $L/libbeat_hip_base.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_tp06_final.txt
timeout -k 10 300 python tools/ab_ode_inproc.py --n 256 --model torord --reps 8 --allocs 1 --dt 0.05 --json gpurun_out/r06_inproc_torord_final.json $L/libbeat_hip_base.so $L/libbeat_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_inproc_torord_final.txt
run() { BEAT_BENCH_BATCHED=0 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(c['ode_ms'],3), 'pde', round(c['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'ode', round(f['ode_ms'],3), '| place', c.get('state_placement')['candidates'], 'frac', round(d['roofline']['frac'],3))"; }
for i in 1 2 3 4; do run cur; done | tee gpurun_out/r06_bench14.txt
shell() { timeout -k 10 300 python tools/bench_biv.py --size $1 --steps 20 2>/dev/null | tail -1; }
for i in 1 2; do shell 400; done | tee gpurun_out/r06_shell14.txt
shell 520 | tee -a gpurun_out/r06_shell14.txt
