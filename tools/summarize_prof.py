#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary.

usage: summarize_prof.py <dir with trace_kernel_stats.csv, pmc_*_counter_collection.csv> <out.md> [title]
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

d = Path(sys.argv[1])
out = Path(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else d.name
OURS = ("ode_step_kernel", "stencil_kernel", "cg_update_kernel", "cg_pupdate_kernel", "reduce_partials_kernel",
        "pcg_next_kernel", "pcg_begin_kernel", "minmax_partial_kernel", "copy", "fill_kernel", "fused", "cg_", "x_flush",
        "var_", "assemble_rows", "dot_partial", "rows_dirichlet", "ode_run_kernel", "gather_kernel", "scatter_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


lines = [f"# {title}", ""]
stats = d / "trace_kernel_stats.csv"
if stats.is_file():
    lines += ["## rocprofv3 --kernel-trace --stats (our kernels)", "",
              "| kernel | calls | avg us | min us | max us | total ms | % of GPU time |", "|---|---:|---:|---:|---:|---:|---:|"]
    with open(stats) as f:
        for r in csv.DictReader(f):
            if not any(k in r["Name"] for k in OURS) or "at::native" in r["Name"]:
                continue
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
    lines.append("")
out.write_text("\n".join(lines))
print("\n".join(lines))
