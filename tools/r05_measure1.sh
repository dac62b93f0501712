#!/bin/bash
# round 5: the headline measurement (bench line + rocprofv3 kernel trace + three PMC passes), then configs[2]
export BEAT_ROUND=r05
bash tools/measure_round.sh 512 && bash tools/measure_round.sh 256iso --size 256 --iso --steps 200 --warmup 20
