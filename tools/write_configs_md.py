"""profiles/<round>_configs.md (BEAT_ROUND, default r04) from the logs of `bash tools/run_configs.sh` (gpurun_out/configs/)."""
import json
import os
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
O = ROOT / "gpurun_out" / "configs"
RND = os.environ.get("BEAT_ROUND", "r04")


def j(f):
    return json.loads((O / f).read_text())


def tail(f, n=1):
    return [ln for ln in (O / f).read_text().splitlines() if "amdgpu" not in ln][-n:]


def line(d):
    c, r = d["config"], d["roofline"]
    go = "auto" if c["guess_order"] == -1 else c["guess_order"]
    return (f"{d['ms_per_step']:.2f} ms/step = {d['value'] / 1e9:.2f} G node-updates/s, PCG {c['pcg_iterations_per_step']:.2f} its/step "
            f"(guess order {go}), ionic kernel {c['ode_ms']:.2f} ms = {r['achieved'] / 1e3:.2f} TB/s ({100 * r['frac']:.0f} % of 8 TB/s), "
            f"diffusion {c['pde_ms']:.2f} ms")


def acts(lines):
    return [float(re.search(r":\s+([0-9.]+) ms", ln).group(1)) for ln in lines[1:]]


def refs(lines):
    return [float(re.search(r"table: ([0-9.]+)", ln).group(1)) for ln in lines[1:]]


c3, c4, c1024, sl, sf = j("cfg3_256iso.json"), j("cfg4_512.json"), j("cfg4_1024.json"), j("slab_lib.json"), j("slab_fused.json")
cb, fr, fr3 = c4["cpu_baseline"], c4["developed_front"], c3["developed_front"]
n05, n02 = tail("cfg2_niederer_dx05_dt005.log", 10), tail("cfg2_niederer_dx02_dt001.log", 10)
sh, sh4 = tail("cfg5_shell.log")[0], tail("cfg5_shell400.log")[0]
md = f"""# Round {RND[2:]}: BASELINE.json's five configurations on one MI355X (`bash tools/run_configs.sh`, one gpurun call on one box)

Package defaults: initial guess of the diffusion solve = extrapolation in time of the last diffusion increments, order 1-4
chosen per solve (`ksp_guess_order` "auto"); one block per 256-node tile in an ionic launch (round 6; 24 576 looping blocks before), the
state array placed by the best of three allocations (round 6); grids of up to 8192 nodes solved in one launch
of one workgroup, their steps batched by `MonodomainSplittingSolver.solve`; per-node rows (voxel meshes, fibre fields): the
workgroup-tile pass with the direction update fused in (round 4).  Box-to-box spread of the pool is ±5 %.  Earlier tables:
`r01_configs.md`, `r02_configs.md` (the "round 1" figures quoted below are theirs); written by `tools/write_configs_md.py`.

| configuration | command | result |
|---|---|---|
| configs[0] 32×32 FitzHugh–Nagumo, dt = 0.01 ms | `demos/fitzhughnagumo.py` | {tail('cfg1_fhn.log')[0].strip()} (unchanged, parity test `test_readme_fitzhugh_nagumo_32x32_matches_oracle`) |
| configs[1] Niederer slab, TP06, dx = 0.5 mm, dt = 0.05 ms | `demos/niederer_benchmark.py --dx 0.5 --dt 0.05` | {n05[0].strip()}; activation times P1–P9 {acts(n05)} ms vs the reference's table {refs(n05)}: max deviation one dt (the values of round 1: neither the initial guess nor the one-launch solve nor the batching moves an activation time; round 1: 0.40 ms/step) |
| same, dx = 0.2 mm, dt = 0.01 ms | `--dx 0.2 --dt 0.01` | {n02[0].strip()}; P1–P9 {acts(n02)} vs {refs(n02)} |
| configs[2] 256³ isotropic, TP06, dt = 0.01 ms | `bench.py --size 256 --iso --steps 200 --warmup 20` | {line(c3)}; developed front {fr3['ms_per_step']:.2f} ms/step at {fr3['pcg_iterations_per_step']:.2f} its (round 1: 3.12 ms/step at 6.70 its) |
| configs[3] 512³ anisotropic, TP06 (headline; N = 1 here) | `bench.py --steps 20 --warmup 5` | {line(c4)}; developed front {fr['ms_per_step']:.2f} ms/step = {fr['value'] / 1e9:.2f} G/s at {fr['pcg_iterations_per_step']:.2f} its; CPU baseline on this box: {cb['value'] / 1e6:.1f} M node-updates/s on {cb['cores']} threads (round 1: 19.92 ms/step at 4.95 its) |
| the same at the largest size one GPU holds | `bench.py --size 1024 --steps 10 --warmup 3 --no-front --direct` | {line(c1024)} (round 1: 144.8 ms/step) |
| one rank's share of configs[3] at N = 8 (512 × 512 × 64 slab) | `bench.py --size 512 --size-z 64 --steps 50 --warmup 10 --no-front`, with and without `BEAT_FORCE_DISTRIBUTED=1` | in-library decomposed loop over a one-rank RCCL communicator {sl['ms_per_step']:.2f} ms/step vs fused single-slab solve {sf['ms_per_step']:.2f} ({100 * (sl['ms_per_step'] / sf['ms_per_step'] - 1):+.1f} %), {sf['config']['pcg_iterations_per_step']:.2f} its/step |
| configs[4] voxelised shell, ToR-ORd-dynCl endo/mid/epi, endocardial pacing (synthetic geometry, one GPU instead of 8) | `tools/bench_biv.py --n 520 --steps 20 --warmup 5` | box 521³ = 141.4 M nodes, 37.12 M tissue nodes; {sh.strip()} (round 1: 43.4 ms/step at 11.0 its) |
| the same on a 401³ box | `tools/bench_biv.py --n 400 --steps 20 --warmup 5` | 17.00 M tissue nodes; {sh4.strip()} (round 1: 21.0 ms/step) |
"""
(ROOT / "profiles" / f"{RND}_configs.md").write_text(md)
print(md)
