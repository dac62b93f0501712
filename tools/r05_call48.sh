#!/bin/bash
# round 5: the right-hand side with shifted SUMS on interior waves (the build) against operand-wise shifts (libbeat_hip_c0.so): tests on the build, bench A/B, kernel times
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
L=$R/fenicsx-beat_amd/beat/lib
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_guess_gpu.py tests/test_properties_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r05_tests48.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r05_tests47.log
[ $rc = 0 ] || exit 1
run() { BEAT_BENCH_BATCHED=0 BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2 3; do
  run build $L/libbeat_hip.so
  run combo0 $L/libbeat_hip_c0.so
done | tee gpurun_out/r05_ab_rr_combo.txt
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0
for v in cur c0; do
  lib=$L/libbeat_hip.so; [ $v = c0 ] && lib=$L/libbeat_hip_c0.so
  rm -rf /tmp/kt_$v
  BEAT_HIP_LIBRARY=$lib timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_$v -o kt --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-front > /tmp/kt_$v.json 2> /tmp/kt_$v.err || echo "trace failed"
  python3 - /tmp/kt_$v $v <<'PY'
import csv, sys, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rr_kernel" in n:
            dur[n[n.index("rr_kernel<") + 10:][:13]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for n, v in sorted(dur.items()):
    v = [x for x in v if x > 50.0]
    if len(v) > 3:
        out.append("%s %.1f" % (n.replace(" ", ""), sum(v) / len(v)))
print(sys.argv[2], " | ".join(out))
PY
done 2>&1 | tee $R/gpurun_out/r05_rr_combo_kernels.txt
