#!/usr/bin/env python3
"""Copies the condensed results of `bash tools/measure_round.sh <tag> [bench args]` (gpurun_out/prof_<round>_<tag>/) into
profiles/<round>_<tag>* (round from BEAT_ROUND, default r04): the summary with a header quoting the bench line of the same box, the PMC json with the
configuration bench.py checks before it quotes `roofline.traffic` / `roofline.valu` from it, the kernel statistics and
the bench line.   usage: copy_round_profiles.py <tag>"""
import json
import os
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
RND = os.environ.get("BEAT_ROUND", "r04")
O, P = ROOT / "gpurun_out" / f"prof_{RND}_{tag}", ROOT / "profiles"
d = json.loads((O / "bench.json").read_text().strip().splitlines()[-1])
fr, r = d.get("developed_front"), d["roofline"]
st = r.get("inplace_stream") or {}
n = round(d["config"]["nodes"] ** (1 / 3))
head = f"""# Round {RND[2:]}: bench.py ({d['config']['workload'].split(',')[0]}, TP06, 1 MI355X) under rocprofv3

Produced by `tools/measure_round.sh {tag}` on one gpurun box: the bench line (`{RND}_{tag}_bench.json`: {d['ms_per_step']:.2f} ms/step =
{d['value'] / 1e9:.2f} G node-updates/s, k = {d['config']['pcg_iterations_per_step']:.2f}"""
if fr:
    head += f"; developed front {fr['ms_per_step']:.2f} ms/step, k = {fr['pcg_iterations_per_step']:.2f}"
rp = st.get("rows_pattern") or {}
head += f"""; ionic kernel
{r['achieved'] / 1e3:.2f} TB/s of algorithmic bytes ({r['bytes_per_node']:.1f} B/node) = {r['frac']:.3f} of 8 TB/s = {st.get('kernel_frac_of_it', float('nan')):.2f} of the {st.get('rate', float('nan')) / 1e3:.2f} TB/s the
library's own in-place streaming probe reached over the same state array in that run, {rp.get('kernel_frac_of_it', float('nan')):.2f} of the
{rp.get('rate', float('nan')) / 1e3:.2f} TB/s its own access pattern -- all state rows at one node index -- streams at without arithmetic; torch's
`x.mul_(1.0)`: {st.get('torch_mul_rate', float('nan')) / 1e3:.2f}), then the same command under `rocprofv3 --kernel-trace
--stats` (10 steps, `{RND}_{tag}_kernel_stats.csv`) and three PMC passes (FETCH_SIZE | WRITE_SIZE | SQ / GRBM counters; 4
steps each), condensed by `tools/summarize_prof.py` (`{RND}_{tag}_pmc.json`).  `stream_kernel` / `rows_kernel` / `mul_` kernels in
the tables are the streaming probes that close every bench run.

Reading the tables: `rr_kernel<MODE, rows, prefetch, guess>`: MODE 0 = PDOT (p = D^-1 r + beta p, p.Ap), 1 = RUPD
(r -= alpha A p), 2 = right-hand side (`true`: with the second register window for the guess increment);
`ode_step_kernel<Model, per-node parameters, pending update, parameter classes>`.  Calls include the latched no-op
launches (min ~5 us).  `fill_kernel` / `copy2_kernel` / the 1-GiB `copyBuffer` calls are the set-up of the state array,
outside the timed steps.
"""
(P / f"{RND}_{tag}.md").write_text(head + (O / "summary.md").read_text().split("\n", 1)[1])
pmc = json.loads((O / "summary.json").read_text())
pmc["config"] = {
    "n": n, "n_gpus": 1, "isotropic": " isotropic slab" in d["config"]["workload"],
    "command": f"bench.py (tools/measure_round.sh {tag}), package defaults: ksp_guess_order auto, 24576 blocks per ionic launch",
    "note": "per-launch means over launches that did real work; HBM bytes: FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE; an ionic-kernel wave walks over ~21 tiles of 64 nodes (valu_instr_per_wave x waves x 64 / nodes = instructions per node)",
}
(P / f"{RND}_{tag}_pmc.json").write_text(json.dumps(pmc, indent=1))
shutil.copy(O / "trace_kernel_stats.csv", P / f"{RND}_{tag}_kernel_stats.csv")
(P / f"{RND}_{tag}_bench.json").write_text(json.dumps(d) + "\n")
print(f"profiles/{RND}_{tag}* refreshed:", d["ms_per_step"], "ms/step")
