#!/bin/bash
# rr kernel durations at several grid sizes (TLB / channel-camping diagnosis)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rr_sizes
mkdir -p $OUT
for n in "$@"; do
  d=$OUT/n$n
  rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o prof --output-format csv -- python3 $ROOT/bench.py --size $n --steps 6 --warmup 2 --cpu-sample 0 --no-front > $d.json 2> $d.err
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $n <<'PY'
import csv,sys
n=int(sys.argv[2]); N=n**3
for r in csv.DictReader(open(sys.argv[1])):
    nm=r['Name']
    for key,b in (('rr_kernel<0',24),('rr_kernel<1',24),('rr_kernel<2',16),('stencil_kernel<1',16),('cg_update_r',24),('cg_pupdate_oop',24),('stencil_kernel<2',24)):
        if key in nm:
            us=float(r['AverageNs'])/1e3
            print(f"n={n:4d} {key:18s} {us:9.1f} us  {N*b/us/1e6:6.2f} TB/s (alg {b} B/node)")
PY
done
