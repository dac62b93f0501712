#!/usr/bin/env python3
"""Copies the condensed results of `bash tools/measure_round2.sh` (gpurun_out/prof_r02/) into profiles/r02_512*: the
summary with a header quoting the bench line of the same box, the PMC json with the configuration bench.py checks before
it quotes `roofline.traffic` / `roofline.valu` from it, the kernel statistics and the bench line."""
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
O, P = ROOT / "gpurun_out" / "prof_r02", ROOT / "profiles"
d = json.loads((O / "bench.json").read_text())
fr, r = d["developed_front"], d["roofline"]
st = r.get("inplace_stream") or {}
head = f"""# Round 2 (final): bench.py (512^3 TP06, 1 MI355X) under rocprofv3

Produced by `tools/measure_round2.sh` on one gpurun box: the default `python bench.py` line (`r02_512_bench.json`: {d['ms_per_step']:.2f} ms/step =
{d['value'] / 1e9:.2f} G node-updates/s, k = {d['config']['pcg_iterations_per_step']:.2f}; developed front {fr['ms_per_step']:.2f} ms/step, k = {fr['pcg_iterations_per_step']:.2f}; ionic kernel
{r['achieved'] / 1e3:.2f} TB/s of algorithmic bytes = {r['frac']:.2f} of 8 TB/s = {st.get('kernel_frac_of_it', float('nan')):.2f} of the {st.get('rate', float('nan')) / 1e3:.2f} TB/s an in-place `x *= 1.0` over the same state
array reached in that run), then the same command under `rocprofv3 --kernel-trace --stats` (10 steps,
`r02_512_kernel_stats.csv`) and three PMC passes (FETCH_SIZE | WRITE_SIZE | SQ / GRBM counters; 4 steps each), condensed
by `tools/summarize_prof.py` (`r02_512_pmc.json`).  The state earlier in the round (x0 = v_, one block per ionic tile:
18.1-18.8 ms/step) is in the history of these files.  `mul_` kernels in the tables are that in-place streaming probe.

Reading the tables: `rr_kernel<MODE, rows, prefetch, guess>`: MODE 0 = PDOT (p = D^-1 r + beta p, p.Ap), 1 = RUPD
(r -= alpha A p), 2 = right-hand side (`true`: with the second register window for the guess increment).  Calls
include the latched no-op launches (min 4.6 us).  `fill_kernel` / `copy2_kernel` / the 1-GiB `copyBuffer` calls are the
set-up of the state array, outside the timed steps.
"""
(P / "r02_512.md").write_text(head + (O / "summary.md").read_text().split("\n", 1)[1])
pmc = json.loads((O / "summary.json").read_text())
pmc["config"] = {
    "n": 512, "n_gpus": 1,
    "command": "bench.py --steps 4..10 --cpu-sample 0 --no-front (tools/measure_round2.sh), package defaults: ksp_guess_order auto, 24576 blocks per ionic launch",
    "note": "per-launch means over launches that did real work; HBM bytes: FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE; an ionic-kernel wave walks over ~21 tiles of 64 nodes (valu_instr_per_wave x waves x 64 / nodes = instructions per node)",
}
(P / "r02_512_pmc.json").write_text(json.dumps(pmc, indent=1))
shutil.copy(O / "trace_kernel_stats.csv", P / "r02_512_kernel_stats.csv")
shutil.copy(O / "bench.json", P / "r02_512_bench.json")
print("profiles/r02_512* refreshed:", d["ms_per_step"], "ms/step")
