#!/bin/bash
# A round's measurement on the GPU box (run through gpurun from the repo root; BEAT_ROUND names the round, default r04):
#   bash tools/measure_round.sh 512                    -> gpurun_out/prof_r04_512/     (BASELINE configs[3], the headline)
#   bash tools/measure_round.sh 256iso --size 256 --iso -> gpurun_out/prof_r04_256iso/  (BASELINE configs[2])
# the bench line, then the SAME bench command under rocprofv3 -- kernel trace + stats, and three PMC passes (FETCH_SIZE |
# WRITE_SIZE | SQ / GRBM counters), each in its own run with the kernel trace only.  Condense with
#   python tools/copy_round_profiles.py <tag>     (-> profiles/r04_<tag>.md, _pmc.json, _kernel_stats.csv, _bench.json)
set -e
TAG=$1
shift
R=$PWD
RND=${BEAT_ROUND:-r04}
O=$R/gpurun_out/prof_${RND}_$TAG
rm -rf $O
mkdir -p $O
timeout -k 10 600 python bench.py "$@" > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o trace --output-format csv -- python3 $R/bench.py "$@" --steps 10 --warmup 3 --cpu-sample 0 --no-front > $O/trace.json 2> $O/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o pmc_fetch --output-format csv -- python3 $R/bench.py "$@" --steps 4 --warmup 1 --cpu-sample 0 --no-front > $O/fetch.json 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o pmc_write --output-format csv -- python3 $R/bench.py "$@" --steps 4 --warmup 1 --cpu-sample 0 --no-front > $O/write.json 2> $O/write.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O -o pmc_sq --output-format csv -- python3 $R/bench.py "$@" --steps 4 --warmup 1 --cpu-sample 0 --no-front > $O/sq.json 2> $O/sq.log
rm -f $O/*kernel_trace.csv $O/*agent_info.csv
cd $R
python3 tools/summarize_prof.py $O $O/summary.md "Round ${RND#r0}: bench.py $* (1 MI355X) under rocprofv3" --json $O/summary.json > /dev/null
rm -f $O/*counter_collection.csv
du -sh $O
