#!/bin/bash
# One pass over BASELINE.json's five configurations on one MI355X (run through gpurun from the repo root); the logs land
# in gpurun_out/configs/ and are condensed into profiles/ by hand.
set -e
O=$PWD/gpurun_out/configs
mkdir -p $O
python demos/fitzhughnagumo.py > $O/cfg1_fhn.log 2>&1
python demos/niederer_benchmark.py --dx 0.5 --dt 0.05 > $O/cfg2_niederer_dx05_dt005.log 2>&1
python demos/niederer_benchmark.py --dx 0.2 --dt 0.01 > $O/cfg2_niederer_dx02_dt001.log 2>&1
python bench.py --size 256 --iso --steps 200 --warmup 20 --cpu-sample 0 > $O/cfg3_256iso.json 2> $O/cfg3.err
python bench.py --steps 20 --warmup 5 > $O/cfg4_512.json 2> $O/cfg4.err
python tools/bench_biv.py --n 520 --steps 10 > $O/cfg5_shell.log 2>&1
for f in cfg1_fhn.log cfg2_niederer_dx05_dt005.log cfg2_niederer_dx02_dt001.log cfg5_shell.log; do echo "== $f"; grep -v amdgpu $O/$f | tail -14; done
