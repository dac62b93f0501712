#!/bin/bash
# round 5: raw-buffer loads in the register-row kernels: none (b0), every mode (b1), every mode but the right-hand side (b2): per-kernel
# times under the kernel trace and the bench's diffusion part, alternating on one box
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
L=$R/fenicsx-beat_amd/beat/lib
cd /tmp && export TMPDIR=/tmp
export BEAT_BENCH_BATCHED=0
for rep in 1 2; do for v in b0 b1 b2; do
  rm -rf /tmp/kt_$v
  BEAT_HIP_LIBRARY=$L/libbeat_hip_$v.so timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_$v -o kt --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --cpu-sample 0 > /tmp/kt_$v.json 2> /tmp/kt_$v.err || echo "trace failed"
  python3 - /tmp/kt_$v $v <<'PY'
import csv, sys, glob, collections, json
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
done; done 2>&1 | tee $R/gpurun_out/r05_rr_buf_kernel_ab.txt
cd $R
run() { BEAT_HIP_LIBRARY=$2 timeout -k 10 240 python bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());f=d['developed_front'];print('$1', round(d['ms_per_step'],3), 'ode', round(d['config']['ode_ms'],3), 'pde', round(d['config']['pde_ms'],3), '| front', round(f['ms_per_step'],3), 'pde', round(f['pde_ms'],3))"; }
for rep in 1 2 3; do for v in b0 b1 b2; do run $v $L/libbeat_hip_$v.so; done; done | tee gpurun_out/r05_ab_rr_buf3.txt
