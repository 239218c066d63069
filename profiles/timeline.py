#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for the LAST n sweep launches (the timed region's
tail), what the GPU did between consecutive sweep kernels: the idle gap on the sweep stream and which other kernels ran
beside / between them.  Used to find the light kernels that are exposed (not hidden behind a sweep).

    python profiles/timeline.py gpurun_out/r03_tl/*/*_kernel_trace.csv [n]
"""
import csv
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"], r["Queue_Id"]) for r in rows]
    ev.sort()
    sweeps = [e for e in ev if "sweep_kernel" in e[2]]
    sweeps = sweeps[-n_last:]
    t0 = sweeps[0][0]
    print("sweep launches: %d, span %.3f ms, busy %.3f ms" % (len(sweeps), (sweeps[-1][1] - t0) / 1e6, sum(e[1] - e[0] for e in sweeps) / 1e6))
    gaps = []
    for a, b in zip(sweeps, sweeps[1:]):
        gap = (b[0] - a[1]) / 1e3
        between = [e for e in ev if e[0] < b[0] and e[1] > a[1] and "sweep_kernel" not in e[2]]
        gaps.append(gap)
        if gap > 20:
            print("gap %8.1f us after sweep ending at %+.3f ms (dur %.3f ms); running in the gap:" % (gap, (a[1] - t0) / 1e6, (a[1] - a[0]) / 1e6))
            for e in between:
                print("      %-48s stream %s queue %s  %+9.1f .. %+9.1f us rel. gap start (dur %.1f us)" % (e[2], e[3], e[4], (e[0] - a[1]) / 1e3, (e[1] - a[1]) / 1e3, (e[1] - e[0]) / 1e3))
    print("sum of gaps %.3f ms over %d sweeps; median %.1f us" % (sum(gaps) / 1e3, len(sweeps), sorted(gaps)[len(gaps) // 2]))
    # per-kernel totals inside the span
    tot = {}
    for e in ev:
        if e[0] >= t0 and e[1] <= sweeps[-1][1]:
            tot.setdefault(e[2], [0, 0])
            tot[e[2]][0] += 1
            tot[e[2]][1] += e[1] - e[0]
    for k, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-48s %5d calls %10.3f ms total %9.1f us avg" % (k, c, ns / 1e6, ns / c / 1e3))


if __name__ == "__main__":
    main()
