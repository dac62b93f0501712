#!/bin/bash
# Counter passes for the TP06 ionic kernel of bench.py (run through gpurun from the repo root): what the waves wait for.
R=$PWD
O=$R/gpurun_out/ode_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $O -o $name --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-front > $O/$name.json 2> $O/$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
cd $R
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/ode_pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ode_step_kernel" not in k: continue
        key = "pend" if "true>" in k.split("Tp06Grl1")[-1][:20] and ", true" in k else "plain"
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
for key in tot:
    print("==", key)
    for c in sorted(tot[key]):
        print(f"  {c:28s} {tot[key][c] / cnt[key][c]:.4e}  (x{cnt[key][c]})")
PY
rm -f $O/*counter_collection.csv $O/*kernel_trace.csv $O/*agent_info.csv
