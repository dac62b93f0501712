#!/bin/bash
# round 5: the decomposed solve left open (beat_pde_solve_dist_begin): the multi-rank GPU tests, then one rank's share of the N = 8
# decomposition through the in-library decomposed loop, BEAT_LAZY_KSP_DIST = 0 | 1 alternating, then its kernel trace
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r05_tests21.log 2>&1; rc=$?; echo "distributed tests rc $rc"; tail -5 gpurun_out/r05_tests21.log
[ $rc = 0 ] || exit 1
export BEAT_BENCH_BATCHED=0 BEAT_FORCE_DISTRIBUTED=1
for rep in 1 2 3; do for lz in 0 1; do
  BEAT_LAZY_KSP_DIST=$lz python3 bench.py --size 512 --size-z 64 --steps 50 --warmup 10 --no-front --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('lazy_dist=$lz', round(d['ms_per_step'],3),'ode',round(d['config']['ode_ms'],3),'pde',round(d['config']['pde_ms'],3),'k',d['config']['pcg_iterations_per_step'])"
done; done | tee gpurun_out/r05_slab64_lazy_dist.txt
cd /tmp && export TMPDIR=/tmp
for lz in 0 1; do
  rm -rf /tmp/tr_$lz
  BEAT_LAZY_KSP_DIST=$lz timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr_$lz -o t --output-format csv -- python3 $R/bench.py --size 512 --size-z 64 --steps 20 --warmup 5 --no-front --cpu-sample 0 > /tmp/tr_$lz.json 2> /tmp/tr_$lz.err || echo "trace failed"
  echo "BEAT_LAZY_KSP_DIST=$lz"; python3 $R/tools/trace_gaps.py /tmp/tr_$lz --last 6
done | tee $R/gpurun_out/r05_slab64_lazy_dist_trace.txt
