#!/bin/bash
# A/B of the per-node SpMV variants on the 401^3 shell (tools/bench_voxel.py), per-kernel times from rocprofv3 --kernel-trace --stats.
# usage (through gpurun, from the repo root): bash tools/vtl_ab.sh "BEAT_VTL=0" "BEAT_VTL=1" ...
R=$PWD
O=$R/gpurun_out/vtl_ab
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
k=0
for v in "$@"; do
  k=$((k+1))
  export $v
  rocprofv3 --kernel-trace --stats -d $O -o run$k --output-format csv -- python3 $R/tools/bench_voxel.py --n ${VTL_AB_N:-400} --reps 6 > $O/run$k.log 2>&1
  for name in $v; do unset ${name%%=*}; done
  echo "== $v" >> $O/summary.txt
  grep "spmv_dot\|theta-step" $O/run$k.log >> $O/summary.txt
  python3 - $O/run${k}_kernel_stats.csv >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("spmv", "reduce_partials", "var_update_r", "var_pupdate", "var_rhs")):
        print(f"   {n.replace('(anonymous namespace)::','').split('(')[0][:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
cat $O/summary.txt
