#!/bin/bash
# PMC passes (one counter group per run, kernel trace only) of the bench's diffusion kernels; prints per-kernel means.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rr_pmc
mkdir -p $OUT
for grp in "$@"; do
  tag=$(echo $grp | tr ' ' '_')
  d=$OUT/$tag
  rm -rf $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-front > $d.json 2> $d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if any(k in n for k in ('rr_kernel','stencil_kernel','cg_update_r','cg_pupdate','ode_step')):
        acc[n[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in acc.items():
    print('  ',n)
    for k,v in c.items():
        print(f"       {k:28s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
