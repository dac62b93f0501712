#!/bin/bash
# round 5, first GPU call: streaming ceiling with the library's own kernels; why the guess helps the slab and not the shell
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/stream_probe.py --json gpurun_out/r05_stream.json > gpurun_out/r05_stream.md 2> gpurun_out/r05_stream.err || echo "stream probe failed"
tail -5 gpurun_out/r05_stream.md
timeout -k 10 240 python tools/guess_probe.py 192 40 0.01 0.1 > gpurun_out/r05_guess_slab_dt01.log 2>&1 || echo "slab probe failed"
tail -2 gpurun_out/r05_guess_slab_dt01.log
timeout -k 10 240 python tools/guess_probe.py 192 40 0.05 0.25 > gpurun_out/r05_guess_slab_dt05.log 2>&1 || echo "slab probe 2 failed"
tail -2 gpurun_out/r05_guess_slab_dt05.log
timeout -k 10 400 python tools/shell_guess_probe.py --size 240 --steps 400 --every 20 > gpurun_out/r05_guess_shell240.log 2>&1 || echo "shell probe failed"
tail -12 gpurun_out/r05_guess_shell240.log
