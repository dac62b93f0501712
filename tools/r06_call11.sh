#!/bin/bash
# round 6: the round's measurement of the headline (bench line, kernel trace + stats, three PMC passes) and of configs[2] (256^3 isotropic)
R=$GRAFT_REPO_ROOT
cd $R
export BEAT_ROUND=r06
bash tools/measure_round.sh 512 2>&1 | tail -3
bash tools/measure_round.sh 256iso --size 256 --iso 2>&1 | tail -3
