#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  O=$R/gpurun_out/prof_vslab$v
  rm -rf $O; mkdir -p $O
  export BEAT_VTL_PDOT_DIST=$v
  rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --slab --reps 6 > $O/log.txt 2>&1
  echo "== BEAT_VTL_PDOT_DIST=$v"
  python3 - "$O/t_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s avg %9.1f us total %8.2f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
