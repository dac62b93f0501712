#!/bin/bash
# round 6, final build: bench.py --gpus 4 as the driver launches it (four ranks sharing this box's GPU over gloo) and --gpus 2 launching its own ranks
set -o pipefail
mkdir -p gpurun_out
BEAT_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 4 --steps 6 --warmup 2 --size 128 --cpu-sample 0 > gpurun_out/r06_rehearsal4.json 2> gpurun_out/r06_rehearsal4.err || { echo "rehearsal failed"; tail -5 gpurun_out/r06_rehearsal4.err; }
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_rehearsal4.json").read().strip().splitlines()[-1])
p = r["multi_rank_parity"]
print("n_gpus", r["n_gpus"], "ms/step", round(r["ms_per_step"], 3), "parity ok", p["ok"], {k: {c: v[c]["max_rel_diff"] for c in v if isinstance(v[c], dict)} for k, v in p.items() if isinstance(v, dict)})
PY
BEAT_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 6 --warmup 2 --size 128 --cpu-sample 0 > gpurun_out/r06_selflaunch2.json 2> gpurun_out/r06_selflaunch2.err || { echo "self-launch failed"; tail -5 gpurun_out/r06_selflaunch2.err; }
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_selflaunch2.json").read().strip().splitlines()[-1])
print("n_gpus", r["n_gpus"], "ms/step", round(r["ms_per_step"], 3), "parity ok", r["multi_rank_parity"]["ok"], "transport", r["config"].get("transport"))
PY
