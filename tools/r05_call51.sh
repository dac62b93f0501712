#!/bin/bash
# round 5: the scalar steps of the PCG (roll behind a residual update, start behind a right-hand side) in the launch that sums the partials
# (BEAT_PCG_FUSE=1, the build) against launches of their own (=0): tests, then 256^3 iso, the 512 x 512 x 64 slab, 512^3, the shell
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_guess_gpu.py tests/test_api_gpu.py tests/test_var_gpu.py -x -q -m gpu > gpurun_out/r05_tests51.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r05_tests51.log
[ $rc = 0 ] || exit 1
export BEAT_BENCH_BATCHED=0
run() { BEAT_PCG_FUSE=$1 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('fuse=$1 [$2]', round(d['ms_per_step'],4), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],4), 'k', d['config']['pcg_iterations_per_step'])"; }
for rep in 1 2 3; do
  for f in 0 1; do run $f "--size 256 --iso --steps 200 --warmup 20"; done
  for f in 0 1; do run $f "--size 512 --size-z 64 --steps 100 --warmup 10"; done
done | tee gpurun_out/r05_ab_pcg_fuse.txt
for f in 0 1 0 1; do run $f "--steps 20 --warmup 5"; done | tee -a gpurun_out/r05_ab_pcg_fuse.txt
for f in 0 1 0 1; do echo -n "fuse=$f shell "; BEAT_PCG_FUSE=$f timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1; done | tee -a gpurun_out/r05_ab_pcg_fuse.txt
