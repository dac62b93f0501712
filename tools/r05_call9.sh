#!/bin/bash
# round 5, ninth GPU call: whole GPU suite, then 8 varying parameter rows on compiled instances (TP06, ToR-ORd at 256^3)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full4.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r05_full4.log
timeout -k 10 300 python tools/bench_param_classes.py --n 256 --model tp06 2>&1 | tail -14 | tee gpurun_out/r05_param_rows_tp06.txt
timeout -k 10 300 python tools/bench_param_classes.py --n 256 --model torord 2>&1 | tail -14 | tee gpurun_out/r05_param_rows_torord.txt
