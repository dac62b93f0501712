#!/bin/bash
# Counters of the per-node SpMV on the 401^3 shell (tools/bench_voxel.py): SQ / GRBM pass and L2 hit / miss pass, each in
# its own rocprofv3 run with the kernel trace only.  Output: gpurun_out/prof_spmv/*.txt
set -e
R=$PWD
O=$R/gpurun_out/prof_spmv
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O -o sq --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $O -o tcc --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/tcc.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O -o sq2 --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/sq2.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $O -o sq3 --output-format csv -- python3 $R/tools/bench_voxel.py --n 400 --reps 4 > $O/sq3.log 2>&1 || true
cd $R
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/prof_spmv/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'var_spmv' in n or 'vtl_spmv' in n or 'var_rhs_kernel' in n or 'var_update_r' in n:
            agg[n.replace('(anonymous namespace)::','').split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    print(f)
    for k, c in agg.items():
        print('  ', k, {n: round(sorted(v)[len(v) // 2], 1) for n, v in c.items()}, 'launches', len(next(iter(c.values()))), '(medians; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950)')
PY
rm -f $O/*counter_collection.csv $O/*kernel_trace.csv
