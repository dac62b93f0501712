#!/bin/bash
# round 6: kernel trace + stats of the 401^3 shell (the build and round 5's library), and the FETCH / WRITE passes of the build
R=$GRAFT_REPO_ROOT
cd $R
L=$R/fenicsx-beat_amd/beat/lib
rm -rf gpurun_out/prof_shell gpurun_out/prof_shell_base
bash tools/profile_shell.sh 400 2>&1 | tail -2
O=$R/gpurun_out/prof_shell_base; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BEAT_HIP_LIBRARY=$L/libbeat_hip_base.so BEAT_STATE_PLACE=1 rocprofv3 --kernel-trace --stats -d $O -o trace --output-format csv -- python3 $R/tools/bench_biv.py --n 400 --steps 20 > $O/trace.log 2>&1
grep -v amdgpu $O/trace.log | tail -1
cd $R
rm -f gpurun_out/prof_shell*/*kernel_trace.csv gpurun_out/prof_shell*/*agent_info.csv gpurun_out/prof_shell*/*counter_collection.csv 2>/dev/null
for d in prof_shell prof_shell_base; do echo "== $d"; python3 - gpurun_out/$d/trace_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
PY
done | tee gpurun_out/r06_shell_kernels.txt
