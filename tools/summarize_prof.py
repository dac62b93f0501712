#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary.

usage: summarize_prof.py <dir with trace_kernel_stats.csv, pmc_*_counter_collection.csv> <out.md> [title] [--json out.json]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request, i.e. exactly half of the bytes of a coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -- the table shows both the raw and the
corrected (x2) read volume.  The correction was checked on minmax_partial_kernel, a pure 8 B/lane
streaming read of a known byte count.
"""
import collections
import csv
import sys
from pathlib import Path

argv = list(sys.argv)
json_out = None
if "--json" in argv:
    k = argv.index("--json")
    json_out = Path(argv[k + 1])
    del argv[k : k + 2]
d = Path(argv[1])
out = Path(argv[2])
title = argv[3] if len(argv) > 3 else d.name
OURS = ("ode_step_kernel", "stencil_kernel", "cg_update_kernel", "cg_pupdate_kernel", "reduce_partials_kernel",
        "pcg_next_kernel", "pcg_begin_kernel", "minmax_partial_kernel", "copy", "fill_kernel", "fused", "cg_", "x_flush",
        "var_", "assemble_rows", "dot_partial", "rows_dirichlet", "ode_run_kernel", "gather_kernel", "scatter_kernel",
        "rr_kernel", "rr_next_kernel", "vtl_spmv")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


summary = {"kernels": {}}


lines = [f"# {title}", ""]
stats = d / "trace_kernel_stats.csv"
if stats.is_file():
    lines += ["## rocprofv3 --kernel-trace --stats (our kernels)", "",
              "| kernel | calls | avg us | min us | max us | total ms | % of GPU time |", "|---|---:|---:|---:|---:|---:|---:|"]
    with open(stats) as f:
        for r in csv.DictReader(f):
            if not any(k in r["Name"] for k in OURS) or "at::native" in r["Name"]:
                continue
            summary["kernels"].setdefault(short(r["Name"]), {})["avg_us"] = float(r["AverageNs"]) / 1e3
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                         f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |")
    lines.append("")
pm = {}
for tag in ("fetch", "write"):
    p = d / f"pmc_{tag}_counter_collection.csv"
    if not p.is_file():
        continue
    agg = collections.defaultdict(list)
    with open(p) as f:
        for r in csv.DictReader(f):
            if "at::native" in r["Kernel_Name"] or not any(k in r["Kernel_Name"] for k in OURS):
                continue
            v = float(r["Counter_Value"])
            agg[short(r["Kernel_Name"])].append(v)
    pm[tag] = agg
if pm:
    lines += ["## PMC passes (separate runs): HBM-side bytes per launch", "",
              "Only launches that did real work are averaged (PCG kernels early-exit after convergence; those launches move < 1 MiB).", "",
              "| kernel | launches | FETCH_SIZE raw GiB | read GiB (x2 gfx950 correction) | WRITE_SIZE GiB |", "|---|---:|---:|---:|---:|"]
    names = sorted(set(pm.get("fetch", {})) | set(pm.get("write", {})))
    for n in names:
        def real(vs):
            vs = [v for v in vs if v > 1024.0] or vs
            return sum(vs) / len(vs) / 1024.0 / 1024.0, len(vs)
        f, nf = real(pm.get("fetch", {}).get(n, [0.0]))
        w, nw = real(pm.get("write", {}).get(n, [0.0]))
        lines.append(f"| `{n}` | {max(nf, nw)} | {f:.3f} | {2*f:.3f} | {w:.3f} |")
        summary["kernels"].setdefault(n, {}).update(hbm_read_bytes=2 * f * 2**30, hbm_write_bytes=w * 2**30)
    lines.append("")
# SQ / GRBM pass: VALU instructions per wave, VALU-busy fraction of the SIMD cycles, effective clock
p = d / "pmc_sq_counter_collection.csv"
if p.is_file():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(p) as f:
        for r in csv.DictReader(f):
            if "at::native" in r["Kernel_Name"] or not any(k in r["Kernel_Name"] for k in OURS):
                continue
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["## PMC pass: SQ / GRBM counters (launches that did real work)", "",
              "VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / "
              "kernel time; fp64-issue time = VALU instructions x 4 cycles / 1024 SIMDs / clock.", "",
              "| kernel | waves | VALU instr / wave | VALU busy | wave cycles parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | GUI cycles per XCD |",
              "|---|---:|---:|---:|---:|---:|"]
    for n, c in sorted(agg.items()):
        waves = c.get("SQ_WAVES", [0.0])
        keep = [i for i, wv in enumerate(waves) if wv > 0] or list(range(len(waves)))
        big = max(c.get("SQ_INSTS_VALU", [0.0]))
        keep = [i for i in keep if c["SQ_INSTS_VALU"][i] > 0.5 * big] or keep  # skip latched no-op launches
        def mean(name):
            v = c.get(name, [])
            v = [v[i] for i in keep if i < len(v)]
            return sum(v) / len(v) if v else float("nan")
        wv, iv, av, gui = mean("SQ_WAVES"), mean("SQ_INSTS_VALU"), mean("SQ_ACTIVE_INST_VALU"), mean("GRBM_GUI_ACTIVE")
        busy = 4.0 * av / (1024.0 * gui / 8.0) if gui else float("nan")
        parked = mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES") if mean("SQ_WAVE_CYCLES") else float("nan")
        lines.append(f"| `{n}` | {wv:.4g} | {iv / wv if wv else float('nan'):.1f} | {busy:.3f} | {parked:.3f} | {gui / 8.0:.4g} |")
        summary["kernels"].setdefault(n, {}).update(waves=wv, valu_instr_per_wave=iv / wv if wv else None, valu_busy=busy,
                                                    wave_cycles_parked=parked, gui_cycles_per_xcd=gui / 8.0)
    lines.append("")
out.write_text("\n".join(lines))
if json_out is not None:
    import json

    json_out.write_text(json.dumps(summary, indent=1))
print("\n".join(lines))
