#!/bin/bash
# round 5: the decomposed loop's TIMED steps (the last five ionic launches of a bench run on a decomposed grid are its communication-profile
# steps, with timing events around every exchange: not those), solve left open or waited for; then a generated model's kernel on its own
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0 BEAT_FORCE_DISTRIBUTED=1 BEAT_BENCH_ALT=0
for lz in 0 1; do
  rm -rf /tmp/tr_$lz
  BEAT_LAZY_KSP_DIST=$lz timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr_$lz -o t --output-format csv -- python3 $R/bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --no-front --cpu-sample 0 > /tmp/tr_$lz.json 2> /tmp/tr_$lz.err || { echo "trace failed"; tail -3 /tmp/tr_$lz.err; }
  echo "== BEAT_LAZY_KSP_DIST=$lz (timed steps)"; python3 $R/tools/trace_gaps.py /tmp/tr_$lz --skip-tail 6 --last 8 --pairs | tail -20
done 2>&1 | tee $R/gpurun_out/r05_slab64_timed_gaps.txt
cd $R
unset BEAT_FORCE_DISTRIBUTED
python3 tools/bench_from_ode.py 2>&1 | tee gpurun_out/r05_from_ode_bench.txt
timeout -k 10 600 python -m pytest tests/test_ode_file_gpu.py -x -q -m gpu 2>&1 | tail -3
