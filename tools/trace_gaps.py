#!/usr/bin/env python3
"""Where is the device idle inside a step?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv), orders the dispatches by start time
and lists, per ionic launch (one per step), the idle time between consecutive kernels up to the next ionic launch: total, largest gap
and which kernels it lies between.     python tools/trace_gaps.py <dir or csv> [--last 12] [--skip-tail N] [--pairs]"""
import csv
import sys
from pathlib import Path


def main():
    src = Path(sys.argv[1])
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 12
    files = [src] if src.is_file() else sorted(src.rglob("*kernel_trace.csv"))
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ion = [i for i, r in enumerate(rows) if r[2].startswith("void ode_step_kernel") or "ode_step_kernel<" in r[2]]
    print(f"{len(rows)} dispatches, {len(ion)} ionic launches")
    if "--skip-tail" in sys.argv:  # drop the last N ionic launches (bench.py's extra steps behind the timed region: communication profile)
        ion = ion[: len(ion) - int(sys.argv[sys.argv.index("--skip-tail") + 1])]
    out = []
    for a, b in zip(ion[:-1], ion[1:]):
        seg = rows[a:b + 1]
        busy = sum(e - s for s, e, _ in seg[:-1])
        span = seg[-1][0] - seg[0][0]
        gaps = [(seg[k + 1][0] - seg[k][1], seg[k][2][:40], seg[k + 1][2][:40]) for k in range(len(seg) - 1)]
        big = max(gaps)
        out.append((span, busy, span - busy, big, len(seg) - 1, seg[0][1] - seg[0][0]))
    for span, busy, idle, big, nk, ode in out[-last:]:
        print(f"step {span / 1e6:7.3f} ms  busy {busy / 1e6:7.3f}  idle {idle / 1e3:7.1f} us in {nk} kernels (ionic {ode / 1e6:6.3f} ms); largest gap {big[0] / 1e3:6.1f} us between {big[1]} -> {big[2]}")
    if "--pairs" in sys.argv and len(ion) > last + 1:
        # which kernel boundaries the idle time sits at: mean gap per (kernel -> next kernel) pair over the last steps
        import collections

        seg = rows[ion[-last - 1]:ion[-1] + 1]
        acc = collections.defaultdict(list)
        for k in range(len(seg) - 1):
            a = seg[k][2].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
            b = seg[k + 1][2].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
            acc[(a, b)].append(seg[k + 1][0] - seg[k][1])
        print(f"gaps by kernel boundary over the last {last} steps (per step: count, mean us, total us)")
        for (a, b), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
            print(f"   {a:34s} -> {b:34s} {len(v) / last:5.2f} x {sum(v) / len(v) / 1e3:6.1f} = {sum(v) / last / 1e3:6.1f} us")
    if out:
        tail = out[-last:]
        print(f"mean over the last {len(tail)}: step {sum(o[0] for o in tail) / len(tail) / 1e6:.3f} ms, idle {sum(o[2] for o in tail) / len(tail) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
