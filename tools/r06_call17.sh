#!/bin/bash
# round 6: the round's records -- BASELINE's configurations (tools/run_configs.sh), then the headline and configs[2] under rocprofv3
R=$GRAFT_REPO_ROOT
cd $R
export BEAT_ROUND=r06
bash tools/run_configs.sh > gpurun_out/r06_configs.log 2>&1; tail -12 gpurun_out/r06_configs.log
bash tools/measure_round.sh 512 2>&1 | tail -2
bash tools/measure_round.sh 256iso --size 256 --iso 2>&1 | tail -2
