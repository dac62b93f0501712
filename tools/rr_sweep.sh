#!/bin/bash
# Sweep of the register-row PCG kernel variants (rows per wave, prefetch distance, blocks per launch) under rocprofv3's
# kernel trace: prints the average duration of the kernels of interest per configuration.  Run on the GPU box from the
# repo root:  bash tools/rr_sweep.sh ry,pd,blocks ...
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rr_sweep
mkdir -p $OUT
for cfg in "$@"; do
  IFS=, read ry pd blocks <<< "$cfg"
  d=$OUT/ry${ry}_pd${pd}_b${blocks}
  rm -rf $d
  BEAT_RR=1 BEAT_RR_RY=$ry BEAT_RR_PD=$pd BEAT_RR_BLOCKS=$blocks rocprofv3 --kernel-trace --stats -d $d -o prof --output-format csv -- python3 $ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-front > $d.json 2> $d.err
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== ry=$ry pd=$pd blocks=$blocks  $(python3 -c "import json;d=json.load(open('$d.json'));print('ms/step',round(d['ms_per_step'],3),'pde',round(d['config']['pde_ms'],3))")"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('rr_kernel','stencil_kernel','cg_','x_flush')):
        print("   avg %9.1f max %9.1f us x%5s  %s" % (float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Calls'], n[:60]))
PY
done
