#!/bin/bash
# round 5, second GPU call: slab guess probes (the bench's regime and the shell's dt / h), the shell at smaller dt, the ionic kernel
# with non-temporal loads / stores on its state rows (A = shipped library)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/guess_probe.py 192 40 0.01 0.1 > gpurun_out/r05_guess_slab_dt01.log 2>&1 || echo "slab probe failed"
tail -2 gpurun_out/r05_guess_slab_dt01.log
timeout -k 10 200 python tools/guess_probe.py 192 40 0.05 0.25 > gpurun_out/r05_guess_slab_dt05.log 2>&1 || echo "slab probe 2 failed"
tail -2 gpurun_out/r05_guess_slab_dt05.log
timeout -k 10 200 python tools/shell_guess_probe.py --size 240 --steps 240 --every 20 --dt 0.01 > gpurun_out/r05_guess_shell240_dt01.log 2>&1 || echo "shell probe dt01 failed"
tail -10 gpurun_out/r05_guess_shell240_dt01.log
timeout -k 10 200 python tools/shell_guess_probe.py --size 240 --steps 240 --every 20 --dt 0.025 > gpurun_out/r05_guess_shell240_dt025.log 2>&1 || echo "shell probe dt025 failed"
tail -10 gpurun_out/r05_guess_shell240_dt025.log
L=$PWD/fenicsx-beat_amd/beat/lib
run() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'k', d['config']['pcg_iterations_per_step'])"; }
for x in A nt3 A nt3 nt1 nt2 A nt3; do
  if [ $x = A ]; then run A libbeat_hip.so; else run $x libbeat_hip_$x.so; fi
done | tee gpurun_out/r05_ab_nt.txt
