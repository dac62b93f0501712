#!/bin/bash
# One pass over BASELINE.json's five configurations on one MI355X (run through gpurun from the repo root); the logs land
# in gpurun_out/configs/ and are condensed into profiles/ by hand.  The first process on a fresh box runs slower than
# every later one (4 ms/step on the shell): the FHN demo takes that place.
set -e
O=$PWD/gpurun_out/configs
mkdir -p $O
python demos/fitzhughnagumo.py > $O/cfg1_fhn.log 2>&1
python demos/niederer_benchmark.py --dx 0.5 --dt 0.05 > $O/cfg2_niederer_dx05_dt005.log 2>&1
python demos/niederer_benchmark.py --dx 0.2 --dt 0.01 > $O/cfg2_niederer_dx02_dt001.log 2>&1
python bench.py --size 256 --iso --steps 200 --warmup 20 --cpu-sample 0 > $O/cfg3_256iso.json 2> $O/cfg3.err
python bench.py --steps 20 --warmup 5 > $O/cfg4_512.json 2> $O/cfg4.err
python bench.py --size 1024 --steps 10 --warmup 3 --cpu-sample 0 --no-front --direct > $O/cfg4_1024.json 2> $O/cfg4_1024.err
python tools/bench_biv.py --n 400 --steps 20 --warmup 5 > $O/cfg5_shell400.log 2>&1
python tools/bench_biv.py --n 520 --steps 20 --warmup 5 > $O/cfg5_shell.log 2>&1
BEAT_FORCE_DISTRIBUTED=1 python bench.py --size 512 --size-z 64 --steps 50 --warmup 10 --cpu-sample 0 --no-front > $O/slab_lib.json 2> $O/slab_lib.err
python bench.py --size 512 --size-z 64 --steps 50 --warmup 10 --cpu-sample 0 --no-front > $O/slab_fused.json 2> $O/slab_fused.err
for f in cfg1_fhn.log cfg2_niederer_dx05_dt005.log cfg2_niederer_dx02_dt001.log cfg5_shell400.log cfg5_shell.log; do echo "== $f"; grep -v amdgpu $O/$f | tail -14; done
for f in cfg3_256iso cfg4_512 cfg4_1024 slab_lib slab_fused; do python - <<PY
import json
d = json.load(open("$O/$f.json")); c = d["config"]; fr = d.get("developed_front")
print("$f", round(d["ms_per_step"], 3), "ms/step", round(d["value"] / 1e9, 3), "G/s k", c["pcg_iterations_per_step"], "ode", round(c["ode_ms"], 3), "pde", round(c["pde_ms"], 3),
      "| front", None if not fr else (round(fr["ms_per_step"], 3), fr["pcg_iterations_per_step"]))
PY
done
