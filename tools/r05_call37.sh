#!/bin/bash
# round 5: the register-row kernels with raw-buffer loads (-DBEAT_RR_BUF=1: no clamped addresses, no zeroing selects; RUPD at 162 VGPRs = 3
# waves instead of 174 = 2) against the build: tests of the diffusion paths on the variant, then bench A/B with the developed front
set -o pipefail
mkdir -p gpurun_out
L=$PWD/fenicsx-beat_amd/beat/lib
BEAT_HIP_LIBRARY=$L/libbeat_hip_buf.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_guess_gpu.py tests/test_properties_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { BEAT_BENCH_BATCHED=0 BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2 3 4; do
  run base $L/libbeat_hip.so
  run buf $L/libbeat_hip_buf.so
done | tee gpurun_out/r05_ab_rr_buf.txt
