#!/bin/bash
# round 5: the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full8.log 2>&1; echo "full rc $?"; tail -5 gpurun_out/r05_full8.log
