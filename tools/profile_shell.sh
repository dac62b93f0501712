#!/bin/bash
# rocprofv3 kernel trace + the FETCH_SIZE / WRITE_SIZE passes of the voxel-shell bench (tools/bench_biv.py); run
# through gpurun from the repo root.  Output: gpurun_out/prof_shell/.
set -e
R=$PWD
O=$R/gpurun_out/prof_shell
N=${1:-400}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o trace --output-format csv -- python3 $R/tools/bench_biv.py --n $N --steps 20 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o pmc_fetch --output-format csv -- python3 $R/tools/bench_biv.py --n $N --steps 3 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o pmc_write --output-format csv -- python3 $R/tools/bench_biv.py --n $N --steps 3 > $O/write.log 2>&1
grep -v amdgpu $O/trace.log | tail -1
