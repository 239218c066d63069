#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints the last `ms` milliseconds of it as a list of kernel executions (start offset,
duration, stream) plus, per kernel name, count / mean / total in that span and the span's busy union -- for pipelines with no
dominant kernel (the reference-default regime after the prefilter), where profiles/timeline.py's sweep-centred view shows nothing.

    python profiles/timeline_kernels.py gpurun_out/r03_tlk/*/*_kernel_trace.csv [ms=40] [min_us=30]
"""
import csv
import sys


def short(n):
    return n.replace("void ", "").split("(")[0][:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
    # the timed region ends with the last final / merge kernel; the profiling pass afterwards is not pipelined: cut at the last
    # prefilter / sweep kernel that is followed within 30 ms by another one
    t_end = ev[-1][1]
    t0 = t_end - int(span_ms * 1e6)
    win = [e for e in ev if e[1] > t0]
    print("span %.1f ms, %d kernel executions" % (span_ms, len(win)))
    tot = {}
    for s, e, n, st in win:
        a = tot.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - max(s, t0)) / 1e3
    edges = sorted([(max(s, t0), 1) for s, e, n, st in win] + [(e, -1) for s, e, n, st in win])
    busy, depth, last = 0, 0, t0
    for t, dlt in edges:
        if depth > 0:
            busy += t - last
        depth += dlt
        last = t
    print("busy union %.2f ms of %.2f" % (busy / 1e6, span_ms))
    for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  %-46s x%-4d mean %9.1f us  total %8.2f ms" % (n, c, us / c, us / 1e3))
    print("executions >= %.0f us:" % min_us)
    for s, e, n, st in win:
        if (e - s) / 1e3 >= min_us:
            print("  %+9.3f ms  %9.1f us  stream %-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, st, n))


main()
