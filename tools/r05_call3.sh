#!/bin/bash
# round 5, third GPU call: new tests (stream probe, multi-rank parity in bench.py), the bench line with the library's own streaming
# ceiling, a longer A/B of non-temporal state rows, the guess probe at the shell's benchmark size
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_distributed_gpu.py -x -q -m gpu -k "stream_probe or four_ranks or exits_nonzero or launches_its_own" > gpurun_out/r05_tests3.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r05_tests3.log
timeout -k 10 300 python bench.py --cpu-sample 0 > gpurun_out/r05_bench3.json 2> gpurun_out/r05_bench3.err || echo "bench failed"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench3.json').read())
print(d['ms_per_step'], d['config']['ode_ms'], d['config']['pde_ms'], json.dumps(d['roofline']['inplace_stream']))
PY
L=$PWD/fenicsx-beat_amd/beat/lib
run() { BEAT_HIP_LIBRARY=$L/$2 timeout -k 10 240 python bench.py --cpu-sample 0 --no-front --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), 'batched', round(d['batched_solve']['ms_per_step'],3), 'bode', round(d['batched_solve']['ode_ms'],3))"; }
for x in A nt3 A nt3 A nt3 A nt3 A nt3 A nt3 A nt3 A nt3; do
  if [ $x = A ]; then run A libbeat_hip.so; else run $x libbeat_hip_$x.so; fi
done | tee gpurun_out/r05_ab_nt2.txt
timeout -k 10 400 python tools/shell_guess_probe.py --size 400 --steps 100 --every 20 > gpurun_out/r05_guess_shell400.log 2>&1 || echo "shell probe failed"
tail -12 gpurun_out/r05_guess_shell400.log
