#!/bin/bash
# L2 hit / miss counts and time of the per-node SpMV on the 401^3 shell for several tile sizes of the tile-ordered
# segment list (BEAT_VAR_TILE; 0 = node order; default 8 since round 3)
R=$PWD
O=$R/gpurun_out/prof_spmv_tiles
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for T in 0 4 8 16; do
  BEAT_VAR_TILE=$T python3 $R/tools/bench_voxel.py --n 400 --reps 6 2>&1 | grep -E "spmv_dot|rhs" | sed "s/^/tile $T: /"
  BEAT_VAR_TILE=$T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $O -o tcc$T --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 3 > $O/tcc$T.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/prof_spmv_tiles/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        if n in ('var_spmv_kernel', 'var_rhs_kernel'):
            agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print(f.split('/')[-1][:6], k, 'req %.1f M  miss %.1f M  hit rate %.2f' % (m['TCC_REQ_sum'] / 1e6, m['TCC_MISS_sum'] / 1e6, m['TCC_HIT_sum'] / m['TCC_REQ_sum']))
PY
rm -f $O/*counter_collection.csv $O/*kernel_trace.csv
