#!/bin/bash
# Measurement round on the GPU box (run through gpurun from the repo root): smoke, GPU tests, bench line, rocprofv3
# kernel trace and the two PMC passes of the same bench command.  Results land in gpurun_out/prof_round/; copy the
# summaries to profiles/ with tools/summarize_prof.py.
set -e
R=$PWD
O=$R/gpurun_out/prof_round
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o pmc_fetch --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o pmc_write --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 > $O/write.log 2>&1
mkdir -p $O/iso256
rocprofv3 --kernel-trace --stats -d $O/iso256 -o trace --output-format csv -- python3 $R/bench.py --size 256 --iso --steps 40 --warmup 5 --cpu-sample 0 > $O/iso256/bench.json 2> $O/iso256/trace.log
ls $O | wc -l
