#!/bin/bash
# round 5, seventh GPU call: the step that leaves its solve open (lazy KSP): tests, then the bench A/B (BEAT_LAZY_KSP=0 against default)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "leaves_its_solve_open or batched_solve or fused or deferred or sparse_rows_run" > gpurun_out/r05_tests7.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r05_tests7.log
run() { BEAT_LAZY_KSP=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];b=d['batched_solve'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'], '| front', round(f['ms_per_step'],3), 'k', f['pcg_iterations_per_step'], '| batched', round(b['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"; }
for x in 0 1 0 1 0 1; do run lazy$x $x; done | tee gpurun_out/r05_ab_lazy.txt
for v in 1 0 1 0; do
  echo "BEAT_LAZY_KSP=$v"; BEAT_LAZY_KSP=$v timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done | tee gpurun_out/r05_biv400_lazy.txt
